#!/usr/bin/env python3
"""bench.py - Chebyshev graph filtering throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
            --master-port P bench.py --gpus N --steps K --warmup W)

A "step" is one pass of the hot path over one batch of synthetic input: an order-30 Heat filter
of 64 signals on a 1M-vertex k=8 sensor graph (~10M stored entries of W) - the north-star
headline configuration (SURVEY.md 8d "NS").  Graph, coefficients, input and output stay resident
in HBM; the timed region is bracketed by barrier + device sync on both sides and the MAX over
ranks is reported.  Each rank owns an independent graph of the same size (weak scaling, no
data-path collective); the RCCL gather of the outputs to rank 0 is timed separately after the
timed region ("gather_ms").

value = n_gpus * N * Nsig * order * steps / time    [vertex.signal.order / s]

roofline: ALGORITHMIC bytes per recurrence-step launch (SURVEY.md 8d:
B_alg = K*(CSR + 3U) + Nf*U, per launch B_alg/K) divided by the average launch duration measured
with HIP events on the engine's own stream over the timed steps.

cpu_baseline: the oracle PORT (numpy/scipy restatement = the reference's algorithm, scipy's
csr_matvecs kernel, single-threaded like the reference) on a bounded column sample of the same
workload, rank 0 at N=1 only; "all_cores" beside it is the same port run by one process per host
core on disjoint signal columns (the reference itself uses one core).

configs: the other BASELINE.json configurations that fit one GPU (configs[1..3] and one rank's share
of configs[4]), each with its own time, algorithmic roofline fraction and parity against oracle
columns, appended after the headline keys (N=1 only; --no-configs skips them).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# What cannot be shown on the driver's GPU box, said in the line itself (VERDICT r5 "Next 7"): /root/reference does not
# travel, so `kind` is "port" there; the reference itself was timed beside the port on a builder box in round 5.
CPU_BASELINE_NOTE = ("kind 'port' = the oracle's restatement of the reference's algorithm (the reference cannot travel to "
                     "this box); the reference itself, pygsp 0.6.1, measured on a builder MI355X box in round 5: "
                     "39.4-41.4 M/s on one core, results identical to the port to 0.0 (profiles/r05_real_pygsp_bench.json)")
REAL_PYGSP_RECORD = ("profiles/r05_real_pygsp_gpu.log: pygsp 0.6.1 and a real MI355X in one process - its own "
                     "pygsp/tests/test_filters.py 25/25 through plugin.install(pygsp), 86 Filter.filter and 38 "
                     "compute_frame calls through the seam (builder box; the reference cannot travel to the driver's)")
HBM_PEAK_GBS = 8000.0       # MI355X HBM3E spec (MI355X_MICROARCH.md)
HBM_COPY_GBS = 6290.0       # measured float4-copy ceiling on this chip (same guide)


def baseline_metric():
    """The metric string of BASELINE.json (falls back to its text if the file is not shipped)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return ("filtered-vertices/sec (N\u00b7Nsig\u00b7K/s) + achieved HBM GB/s vs roofline, "
                "1M-vertex K=30")


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--dtype", choices=["f64", "f32"], default="f64",
                   help="arithmetic type (the reference computes in f64)")
    p.add_argument("--vertices", dest="n", type=int, default=1000000)
    p.add_argument("--knn", type=int, default=8)
    p.add_argument("--nsig", type=int, default=64)
    p.add_argument("--order", type=int, default=30)
    p.add_argument("--scale", type=float, default=50.0)
    p.add_argument("--cpu-cols", type=int, default=16, help="columns of the CPU-baseline sample")
    p.add_argument("--no-cpu", action="store_true")
    p.add_argument("--no-gather", action="store_true")
    p.add_argument("--no-newton", action="store_true")
    p.add_argument("--no-mix", action="store_true", help="skip the mix-ceiling calibration (gspx_bench_step_mix)")
    p.add_argument("--tune-candidates", type=int, default=32,
                   help="set-up: physical backings of the streamed workspaces drawn by DeviceGraph.tune_placement, the "
                        "fastest kept (0: none; profiles/r06_placement.md)")
    p.add_argument("--tune-stride-mb", type=int, default=8000,
                   help="set-up: device memory held between two draws of the placement tuning, so that the candidates "
                        "sample the card's memory at that stride (fast and slow pages come in zones of tens of GB)")
    p.add_argument("--no-e2e", action="store_true", help="skip the numpy-in/numpy-out leg (profiling passes)")
    p.add_argument("--no-configs", action="store_true", help="skip the other BASELINE configs")
    p.add_argument("--no-f32", action="store_true", help="skip the float32 run of the headline workload")
    p.add_argument("--no-chain", action="store_true", help="skip the three-filter device-resident chain")
    p.add_argument("--no-headline", action="store_true",
                   help="profiling passes of one config: skip the headline workload entirely")
    p.add_argument("--only-config", default=None, help="run only the config whose key starts with this (c1..c4)")
    p.add_argument("--config-reps", type=int, default=3, help="timed calls per config (best is reported)")
    p.add_argument("--config-oracle-cols", type=int, default=2, help="oracle columns per config (0: no parity leg)")
    p.add_argument("--calibrate-copy", action="store_true",
                   help="also run the engine's streaming copy kernel once (byte-counter calibration for rocprofv3)")
    p.add_argument("--no-live-traffic", action="store_true",
                   help="do not re-run one step under rocprofv3 --pmc to measure roofline.traffic in this run")
    p.add_argument("--cpu-all-cores", type=int, default=-1,
                   help="processes of the multi-core CPU baseline (-1: one per host core, at most 64; 0: skip)")
    p.add_argument("--tiles", choices=["auto", "off"], default="auto",
                   help="auto = the product default (LDS-staged recurrence step when the graph's order "
                        "is local); off = the plain gather kernels")
    p.add_argument("--opt", action="append", default=[], help="engine option key=value")
    p.add_argument("--reorder", default="auto")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend for N>1 (nccl = RCCL)")
    p.add_argument("--devices", default=None,
                   help="single-process launch (no WORLD_SIZE in the environment): comma-separated GPU ids, one "
                        "driver thread and one libgspx context per entry (default 0..N-1; an id may repeat - "
                        "the one-GPU test hook)")
    p.add_argument("--evaluation", choices=["recurrence", "newton"], default="recurrence",
                   help="recurrence = the reference's three-term Chebyshev recurrence (headline); "
                        "newton = same polynomial, Newton form (extra line 'newton_form')")
    return p.parse_args()


def smi_sample(device=0):
    """What rocm-smi says about a device right now (called from a thread while the recurrence runs): HBM and junction
    temperatures, clocks, package power.  The memory temperature is the one quantity that moved with the step's
    speed over successive runs on one GPU in round 6 (68 C: 0.612 of 8 TB/s ... 72 C: 0.598, the mix ceiling falling
    with it - profiles/r06_placement.md).  {} when rocm-smi is missing or fails."""
    import re
    import shutil
    import subprocess
    exe = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    try:
        txt = subprocess.run([exe, "-d", str(int(device)), "--showtemp", "--showclocks", "--showpower"], capture_output=True, text=True,
                             timeout=30).stdout
    except Exception:
        return {}
    out = {}
    try:  # which card this is (the population of cards differs: memory vendor, VBIOS; profiles/r06_placement.md)
        ident = subprocess.run([exe, "-d", str(int(device)), "--showuniqueid", "--showmemvendor", "--showvbios"],
                               capture_output=True, text=True, timeout=30).stdout
        for key, pat in (("gpu_unique_id", r"Unique ID:\s*(\S+)"), ("memory_vendor", r"memory vendor:\s*(\S+)"),
                         ("vbios", r"VBIOS version:\s*(\S+)")):
            m = re.search(pat, ident)
            if m:
                out[key] = m.group(1)
    except Exception:
        pass
    for key, pat in (("hbm_temperature_C", r"Sensor memory\) \(C\):\s*([0-9.]+)"),
                     ("junction_temperature_C", r"Sensor junction\) \(C\):\s*([0-9.]+)"),
                     ("sclk_MHz", r"sclk clock level:[^(]*\((\d+)Mhz\)"), ("mclk_MHz", r"mclk clock level:[^(]*\((\d+)Mhz\)"),
                     ("fclk_MHz", r"fclk clock level:[^(]*\((\d+)Mhz\)"), ("power_W", r"Power \(W\):\s*([0-9.]+)")):
        m = re.search(pat, txt)
        if m:
            out[key] = float(m.group(1))
    return out


# ---- HBM bytes per launch, measured in THIS run ---------------------------------------------------
def live_traffic(dtype_flag, timeout_s=150, device=0):
    """roofline.traffic measured now, on this box: one call of the headline workload re-run under
    `rocprofv3 --pmc FETCH_SIZE` and again under `--pmc WRITE_SIZE` (separate passes, no trace domain - the
    HBM section of MI355X_MICROARCH.md: FETCH_SIZE costs 3 of the 4 TCC slots and counts half the bytes of
    wide reads on gfx950, hence 2 x), summed over the recurrence-step launches.  Returns
    (bytes per launch, launches, calibration) or None when rocprofv3 is missing / fails / times out."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None
    child = [sys.executable, os.path.abspath(__file__), "--steps", "1", "--warmup", "0", "--dtype", dtype_flag,
             "--no-cpu", "--no-newton", "--no-mix", "--no-e2e", "--no-configs", "--no-live-traffic", "--no-f32",
             "--calibrate-copy", "--tune-candidates", "0"]
    # a single-process child on ONE GPU (this rank's), whatever launched the parent
    child_env = {k: v for k, v in os.environ.items()
                 if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK",
                              "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
    child_env["TMPDIR"] = "/tmp"
    if device:
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        ids = vis.split(",") if vis else None
        child_env["HIP_VISIBLE_DEVICES"] = ids[device] if ids and device < len(ids) else str(device)
    got = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        out = tempfile.mkdtemp(prefix="gspx_pmc_", dir="/tmp")
        try:
            subprocess.run([prof, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--"] + child,
                           cwd="/tmp", env=child_env, stdout=subprocess.DEVNULL,
                           stderr=subprocess.DEVNULL, timeout=timeout_s, check=True)
            step, copy = [], []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if r.get("Counter_Name") != counter:
                        continue
                    (step if "k_step" in r["Kernel_Name"] else copy if "k_permute_in" in r["Kernel_Name"] else []
                     ).append(float(r["Counter_Value"]))
            got[counter] = (step, copy)
        except Exception:
            return None
        finally:
            shutil.rmtree(out, ignore_errors=True)
    fs, ws = got["FETCH_SIZE"][0], got["WRITE_SIZE"][0]
    n = min(len(fs), len(ws))
    if n == 0:
        return None
    cal = None
    if got["FETCH_SIZE"][1] and got["WRITE_SIZE"][1]:
        # the child's last launch of the engine's copy kernel (k_permute_in) reads and writes 1 GiB: WRITE_SIZE
        # counts it in full, FETCH_SIZE counts half - the factor 2 above, re-checked in the same passes
        cal = {"copy_kernel_fetch_KiB": got["FETCH_SIZE"][1][-1], "copy_kernel_write_KiB": got["WRITE_SIZE"][1][-1],
               "copy_kernel_bytes_each_way_KiB": 1 << 20}
    return (2.0 * sum(fs[:n]) + sum(ws[:n])) * 1024.0 / n, n, cal


# ---- CPU baseline on all host cores ---------------------------------------------------------------
_POOL = {}


def _pool_column(j):
    """One worker = one process = one signal column through the oracle port (fork: L is shared)."""
    from oracle import cheby_oracle as orc
    L, lmax, c, x = _POOL["L"], _POOL["lmax"], _POOL["c"], _POOL["x"]
    y = orc.cheby_op(L, lmax, c, x[:, j:j + 1])
    return float(np.abs(y).max())


def cpu_all_cores(N, knn, K, scale, procs):
    """The oracle port on every host core at once: `procs` processes, one signal column each, same
    graph / coefficients as the headline (built on the host, before any HIP call in this process - the
    workers are forked).  Returns the cpu_baseline.all_cores object."""
    import multiprocessing as mp

    from oracle import cheby_oracle as orc
    from pygsp_amd import graphs
    W, _ = graphs.sensor_weights(N, k=knn, seed=42)
    L = orc.laplacian(W).astype(np.float64)
    lmax = 2.0 * float(np.ravel(W.sum(axis=0)).max())  # an upper bound (graph.py:947), as 'bounds' gives
    c = orc.compute_cheby_coeff(orc.heat_kernel(scale, lmax), lmax, K)
    x = np.random.default_rng(7).standard_normal((N, procs))
    _POOL.update(L=L, lmax=lmax, c=c, x=x)
    with mp.get_context("fork").Pool(procs) as pool:
        pool.map(_pool_column, range(min(procs, 4)))  # warm-up: page in, spin the workers up
        t0 = time.perf_counter()
        pool.map(_pool_column, range(procs), chunksize=1)
        dt = time.perf_counter() - t0
    _POOL.clear()
    return {"value": N * procs * K / dt, "unit": "vertex*signal*order/s", "cores": procs, "kind": "port",
            "sample": "same graph/coefficients, {} signal columns, one per process ({} host cores present), "
                      "order {}, float64, {:.1f} s".format(procs, os.cpu_count(), K, dt)}


# ---- the other BASELINE.json configs --------------------------------------------------------------
def gather_ceiling(ctx, N, row_bytes, entries, blocks=1, p_intra=0.0):
    """The rate this box serves random row gathers of this shape at, measured now (gspx_bench_gather: rows of
    `row_bytes` fetched by 32-bit indices from an N-row panel, `entries` gathers, nothing else in the kernel):
    best over the in-flight depth and the workgroups per CU.  Returns the roofline `peak` object."""
    best = None
    for wg in (4, 8):
        for depth in (4, 8, 16):
            try:
                ms, gbps = ctx.bench_gather(N, row_bytes, entries, depth, blocks, p_intra, wg, 3)
            except Exception:
                continue
            if best is None or gbps > best["GBps"]:
                best = {"GBps": gbps, "ms": ms, "in_flight": depth, "workgroups_per_cu": wg}
    return best


def run_config(key, workload, G, bank, nsig, K, dtype, ctx, oracle_cols=2, reps=3, sbm=None):  # noqa: C901
    """One BASELINE config, device resident: best-of-`reps` device time of the whole call (HIP events),
    algorithmic bytes B_alg = K (CSR + 3U) + Nf U, parity of `oracle_cols` columns
    against the oracle (its own Laplacian from the same W)."""
    from oracle import cheby_oracle as orc
    from pygsp_amd import filters
    elt = np.dtype(dtype).itemsize
    G.estimate_lmax("bounds")
    lmax = float(G.lmax)
    c = np.atleast_2d(np.array(filters.compute_cheby_coeff(bank, m=K)))
    Nf, N = c.shape[0], G.N
    dev = G.device_graph()
    x = np.random.default_rng(0).standard_normal((N, nsig)).astype(dtype)
    bx, by = ctx.upload(x), ctx.alloc(x.nbytes * Nf)
    best, tm = None, None
    for _ in range(reps + 1):  # the first call allocates the workspace
        ms = dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, lmax)
        if best is None or ms < best:
            best, tm = ms, ctx.last_timing()
    y = by.download((Nf, N, nsig), dtype)
    bx.free()
    by.free()
    cols = list(range(min(oracle_cols, nsig)))
    err = None
    if cols:
        L = orc.laplacian(G.W.astype(np.float64), G.lap_type)
        ref = orc.cheby_op(L, lmax, c, x[:, cols].astype(np.float64)).reshape(Nf, N, len(cols))
        err = float(np.max(np.abs(y[:, :, cols] - ref)) / np.max(np.abs(ref)))
    nnz_l, nnz_int = dev.nnz_l, dev.nnz_internal
    U = N * nsig * elt
    csr = nnz_l * (elt + 4) + 4 * (N + 1)
    b_alg = K * (csr + 3 * U) + Nf * U
    step_ms = tm["steps_ms"] / max(tm["step_launches"], 1)
    gather_bytes = nnz_l * nsig * elt  # one panel row per stored entry and recurrence step
    tiled = bool(G.tile_stats and G.tile_stats.get("enabled"))
    # graphs without vertex locality: the yardstick is the rate of random row gathers, measured on this box in
    # this run with the same row width, panel and entry count (uniform indices for ER; block-local ones, one XCD
    # per pair of blocks, for the SBM)
    roof_gather = None
    if not tiled and nsig * elt in (64, 128, 256, 512) and N >= 100000:
        blocks, p_intra = sbm if sbm else (1, 0.0)
        peak = gather_ceiling(ctx, N, nsig * elt, int(nnz_l), blocks, p_intra)
        if peak:
            rate = gather_bytes / (step_ms * 1e-3) / 1e9
            # a step also streams the matrix entries and T_{k-2} in and T_k out: CSR + 2U at the copy rate
            stream_bytes = csr + 2 * U
            copy_rate = None
            try:
                copy_rate = ctx.bench_copy(1 << 29, 3)
            except Exception:
                pass
            # bound of a step: every byte it must pull through the L2-miss path - one line-granular row per stored
            # entry plus the streamed CSR + 2U - at the rate the pure-gather kernel gets lines out of that path
            line = 128
            gather_lines = nnz_l * max(nsig * elt, line)
            bound_ms = peak["ms"] * (gather_lines + stream_bytes) / gather_lines
            roof_gather = {"bound": "l2-miss-gather", "achieved": (gather_lines + stream_bytes) / (step_ms * 1e-3) / 1e9,
                           "peak": gather_lines / (peak["ms"] * 1e-3) / 1e9, "unit": "GB/s (128-byte lines)",
                           # (a MODEL of a step, not a bound: the step's gathers overlap its streams and part of them
                           # hit in the L2, so a step can beat the pure-gather kernel's time - reported capped at 1)
                           "frac": min(1.0, bound_ms / step_ms), "model_over_step": bound_ms / step_ms,
                           "bound_ms": bound_ms, "step_ms": step_ms,
                           "pure_gather": {"ms": peak["ms"], "row_GBps": peak["GBps"], "in_flight": peak["in_flight"],
                                           "workgroups_per_cu": peak["workgroups_per_cu"],
                                           "step_row_GBps": rate, "frac_of_rows_alone": rate / peak["GBps"]},
                           "index_distribution": "uniform" if blocks == 1 else
                           "block-local: {} blocks, p_intra {:.3f}".format(blocks, p_intra),
                           "streamed_bytes_beside_the_gathers": stream_bytes, "copy_GBps": copy_rate,
                           "note": "peak = gspx_bench_gather on this box in this run: the same number of row gathers of "
                                   "the same width from a panel of the same size, no matrix, no FMA, no writes (best "
                                   "over in-flight depth x workgroups per CU).  A recurrence step additionally streams "
                                   "CSR + 2U (entries, T_{k-2} in, T_k out); bound_ms charges those bytes at the same "
                                   "line rate.  frac = min(1, bound_ms / step_ms); model_over_step is the uncapped ratio (> 1 means the "
                                   "step hid part of its streams behind its gathers / hit in the L2)"}
    return {
        "key": key, "workload": workload, "dtype": "f64" if elt == 8 else "f32", "N": N, "nnz_L": int(nnz_l),
        "nnz_internal": int(nnz_int), "Nsig": nsig, "Nf": Nf, "order": K, "lap_type": G.lap_type,
        "ms": best, "value": N * nsig * K / (best * 1e-3), "unit": "vertex*signal*order/s",
        "steps_ms": tm["steps_ms"], "combine_ms": tm["combine_ms"], "permute_ms": tm["permute_ms"],
        "step_launches": tm["step_launches"], "avg_step_ms": step_ms,
        "roofline": dict({"bound": "hbm", "achieved": b_alg / (best * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": b_alg / (best * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_call": b_alg},
                         **config_traffic(key, "f64" if elt == 8 else "f32")),
        # what a graph without vertex locality is really bound by: every stored entry fetches one panel
        # row through the L2 -> Infinity Cache / HBM path
        "gather": {"bytes_per_step": gather_bytes, "rate_GBps": gather_bytes / (step_ms * 1e-3) / 1e9},
        "roofline_gather": roof_gather,
        "kernel": ("k_step_narrow (sub-wave rows: a single signal on a matrix that stays in the L2s)"
                   if tiled and nsig == 1 and dev.nnz_internal * (elt + 4) < (20 << 20) else
                   "k_step_tile (LDS-staged gathers)" if tiled else "plain gather kernels (no vertex locality: no tiles)"),
        "internal_order": "curve / RCM" if G._internal_order() is not None else "none (graph's own order)",
        "parity_vs_oracle": {"max_rel_err": err, "columns": len(cols), "tolerance": 1e-5 if elt == 8 else 1e-3},
        "lmax": lmax,
    }


def run_configs(ctx, only=None, reps=3, oracle_cols=2):
    """BASELINE.json configs[1..4] on one GPU.  ER / SBM graphs come from the device sampler (equal in
    distribution to the reference's constructors); parity is always against the oracle on
    the same W."""
    from pygsp_amd import filters, graphs
    res = []

    def want(k):
        return only is None or k.startswith(only)

    if want("c1"):
        G = graphs.Sensor(100000, seed=42, compute_dtype=np.float64)
        G.estimate_lmax("bounds")  # before the bank is designed: the kernels read / capture G.lmax
        res.append(run_config("c1", "configs[1]: Sensor(N=100000) combinatorial, Heat(50) order 30, 1 signal, f64 "
                              "(cache-resident latency case; replayed as one hipGraph)", G, filters.Heat(G, 50), 1, 30,
                              np.float64, ctx, oracle_cols=min(oracle_cols, 1), reps=max(reps, 12)))
        del G
    if want("c2"):
        N = 1000000
        G = graphs.ErdosRenyi(N, p=10.0 / N, seed=0, compute_dtype=np.float32)
        G.estimate_lmax("bounds")
        res.append(run_config("c2", "configs[2]: ErdosRenyi(N=1000000, p=1e-5), MexicanHat filterbank (6 filters) "
                              "order 50, 64 signals, f32", G, filters.MexicanHat(G, Nf=6), 64, 50, np.float32, ctx,
                              oracle_cols, reps))
        del G
    if want("c3"):
        for dt in (np.float64, np.float32):
            G = graphs.StochasticBlockModel(2000000, k=16, p=9.6e-5, q=2.13e-6, seed=0, lap_type="normalized",
                                            compute_dtype=dt)
            G.estimate_lmax("bounds")
            p_in, q_out, kb = 9.6e-5, 2.13e-6, 16
            intra = p_in / kb / (p_in / kb + q_out * (kb - 1) / kb)  # share of a row's neighbours in its own block
            res.append(run_config("c3", "configs[3]: StochasticBlockModel(N=2000000, k=16, p=9.6e-5, q=2.13e-6) "
                                  "normalized Laplacian, Heat(10) order 30, 16 signals", G, filters.Heat(G, 10), 16, 30,
                                  dt, ctx, oracle_cols, reps, sbm=(kb, intra)))
            del G
    if want("c4"):
        for dt in (np.float64, np.float32):
            G = graphs.Sensor(500000, seed=0, compute_dtype=dt)
            G.estimate_lmax("bounds")
            res.append(run_config("c4", "configs[4], one rank's share: Sensor(N=500000), Heat(50) order 30, 32 signals "
                                  "(the batch of 8 such graphs is one per GPU; bench.py --gpus N is its scaling run)",
                                  G, filters.Heat(G, 50), 32, 30, dt, ctx, oracle_cols, reps))
            del G
    return res


def run_batch_config(local, rank, world, ctx, gdist, rdev, fence, comm):
    """BASELINE.json configs[4]: a batch of 8 independent Sensor(N=500000) graphs x 32 signals, Heat order
    30, sharded over the ranks (8 / world graphs each, no data-path collective); one gather of the 8 output
    blocks to rank 0 at the end (in-library RCCL when the run has a communicator).  Strong scaling of a
    fixed batch: `bench.py --gpus N` for N = 1, 2, 4, 8 gives its curve.  Timed like the headline: barrier +
    device sync on both sides, MAX over ranks; one column of every rank's first graph is checked against
    the oracle."""
    from oracle import cheby_oracle as orc
    from pygsp_amd import filters, graphs
    n_graphs, N5, nsig, K = 8, 500000, 32, 30
    mine = list(gdist.shard_units(n_graphs, rank, world))
    block = N5 * nsig * 8
    by_all = ctx.alloc(max(len(mine), 1) * block)
    jobs = []
    for slot, g in enumerate(mine):
        G = graphs.Sensor(N5, seed=g, compute_dtype=np.float64, device=local)
        G.estimate_lmax("bounds")
        c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=K))
        x = np.random.default_rng(100 + g).standard_normal((N5, nsig))
        jobs.append((G, c, x, ctx.upload(x), by_all.ptr + slot * block))

    def run_all():
        for G, c, _, bx, y_ptr in jobs:
            G.device_graph().cheby_filter_dev(c, bx.ptr, y_ptr, nsig, float(G.lmax))

    run_all()  # warm-up: workspaces, tiles
    fence()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        run_all()
    fence()
    wall = gdist.max_over_ranks((time.perf_counter() - t0) / reps, rdev)
    gather_ms = None
    if comm is not None:
        table = [len(list(gdist.shard_units(n_graphs, r, world))) * block for r in range(world)]
        root = ctx.alloc(sum(table)) if rank == 0 else None
        fence()
        tg = time.perf_counter()
        comm.gather(by_all.ptr, table, 0, root.ptr if rank == 0 else None)
        fence()
        gather_ms = gdist.max_over_ranks((time.perf_counter() - tg) * 1e3, rdev)
        if rank == 0:
            root.free()
    err = 0.0
    if jobs:
        G, c, x, _, _ = jobs[0]
        y = by_all.download((len(mine), N5, nsig), np.float64)[0]
        ref = orc.cheby_op(orc.laplacian(G.W), float(G.lmax), c[0], x[:, :1])
        err = float(np.max(np.abs(y[:, :1] - ref.reshape(N5, 1))) / np.max(np.abs(ref)))
    err = gdist.max_over_ranks(err, rdev)
    for G, _, _, bx, _ in jobs:
        bx.free()
        for g_ in list(G._dev.values()):
            g_.destroy()
        G._dev = {}
    by_all.free()
    return {"workload": "configs[4]: 8 x Sensor(N=500000), Heat(50) order 30, 32 signals each, f64, sharded {} per rank "
                        "(strong scaling of a fixed batch)".format("/".join(str(len(list(gdist.shard_units(n_graphs, r, world))))
                                                                             for r in range(world))),
            "n_graphs": n_graphs, "n_gpus": world, "ms": wall * 1e3, "value": n_graphs * N5 * nsig * K / wall,
            "unit": "vertex*signal*order/s", "gather_ms": gather_ms,
            "parity_vs_oracle": {"max_rel_err": err, "columns": 1, "tolerance": 1e-5}}


def kernel_sources_sha():
    """sha256 (16 hex digits) over the HIP sources: what a recorded PMC measurement is stamped with, so that a record
    taken on other kernels is recognised as stale."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "pygsp_amd", "csrc", "*.hip*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def config_traffic(key, dtype):
    """RECORDED fabric traffic of a config's recurrence step: the committed PMC passes of an earlier run
    (profiles/traffic_configs.json, tools/config_traffic.py: separate rocprofv3 --pmc runs of `bench.py --no-headline
    --only-config <key>`; counters cannot be read from inside this process).  Not measured by this run, hence the
    `traffic_recorded*` names beside the live `achieved` / `frac`, the stamp of the run they come from, and
    `traffic_recorded_stale` when the HIP sources have changed since; `traffic` itself stays null.
    The x2 on FETCH_SIZE was calibrated on gathers of unique rows in round 6: every row of <= 128 bytes costs one
    128-byte request, counted as 64 (profiles/r06_gather_calibration.md)."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "traffic_configs.json")))
        t = rec["{}_{}".format(key, dtype)]
    except Exception:
        return {"traffic": None}
    meta = rec.get("_meta") or {}
    return {"traffic": None, "traffic_recorded": t["hbm_bytes_per_step_launch"],
            "traffic_recorded_over_algorithmic": t["traffic_over_algorithmic"],
            "traffic_recorded_TBps": t["hbm_TBps"], "tcc_hit_rate_recorded": t["tcc_hit_rate"],
            "traffic_recorded_stamp": meta or None,
            "traffic_recorded_stale": (meta.get("kernel_sources_sha") != kernel_sources_sha()) if meta else None,
            "traffic_source": "profiles/traffic_configs.json (recorded, not of this run): " + t["method"]}


def reference_cpu_baseline(W, lmax, scale, K, xs):
    """The reference's own Filter.filter on the CPU (filter.py:146-328 -> approximations.py:58-114) for the sample `xs`,
    when a real pygsp is importable here ($PYGSP_PATH first, then an installed package): (seconds, result,
    (version, location)) or None.  Never the product path: the plugin is not installed in this process."""
    import importlib
    extra = os.environ.get("PYGSP_PATH")
    added = False
    if extra and os.path.isdir(os.path.join(extra, "pygsp")) and extra not in sys.path:
        sys.path.insert(0, extra)
        added = True
    try:
        pygsp = importlib.import_module("pygsp")
        approx = pygsp.filters.approximations
        if getattr(approx.cheby_op, "__module__", "").startswith("pygsp_amd"):
            return None  # patched: that would time the device, not the reference
        G = pygsp.graphs.Graph(W)
        G._lmax = lmax
        G._lmax_method = "bounds"
        flt = pygsp.filters.Heat(G, scale)
        flt.filter(xs[:, :1], method="chebyshev", order=K)  # warm-up (builds G.L)
        t0 = time.perf_counter()
        y = flt.filter(xs, method="chebyshev", order=K)
        dt = time.perf_counter() - t0
        return dt, np.asarray(y).reshape(xs.shape), (getattr(pygsp, "__version__", "?"), os.path.dirname(pygsp.__file__))
    except Exception as e:
        sys.stderr.write("bench.py: no reference CPU baseline ({!r})\n".format(e))
        return None
    finally:
        if added:
            sys.path.remove(extra)


def headline_other_dtype(a, ctx, G, c, x, lmax, dtype, oracle=True):
    """The headline workload (same graph, coefficients and signals) in `dtype`: device-resident rate, roofline of the
    recurrence step from its HIP-event times, parity of 2 columns against the float64 oracle."""
    N, nsig, K = a.n, a.nsig, a.order
    elt = np.dtype(dtype).itemsize
    dev = G.device_graph(dtype)
    xs = x.astype(dtype)
    bx, by = ctx.upload(xs), ctx.alloc(xs.nbytes)
    try:
        for _ in range(a.warmup):
            dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, lmax)
        ctx.sync()
        t0 = time.perf_counter()
        steps_ms, launches = 0.0, 0
        for _ in range(a.steps):
            dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, lmax)
            t = ctx.last_timing()
            steps_ms += t["steps_ms"]
            launches += t["step_launches"]
        ctx.sync()
        elapsed = time.perf_counter() - t0
        y = by.download(xs.shape, dtype)[:, :2]
        # the same call as evaluation='auto' would run it (Newton form when the guard clears it), and the mix ceiling
        from pygsp_amd import filters
        newton_ms = y_newton = None
        auto = filters.choose_evaluation("auto", np.atleast_2d(c[0]), dtype, N, nsig)
        if auto in ("newton", "product") and not a.no_newton:
            if auto == "newton":
                nodes, dcoef = filters.cheb_to_newton(c[0])
                run_auto = lambda: dev.newton_filter_dev(nodes, dcoef, bx.ptr, by.ptr, nsig, lmax)  # noqa: E731
            else:
                program = filters.cheb_to_product(c[0], dtype)
                run_auto = lambda: dev.program_filter_dev(program, bx.ptr, by.ptr, nsig, lmax)  # noqa: E731
            run_auto()
            n_ms = 0.0
            for _ in range(max(3, a.steps // 2)):
                run_auto()
                n_ms += ctx.last_timing()["steps_ms"]
            newton_ms = n_ms / (max(3, a.steps // 2) * K)
            y_newton = by.download(xs.shape, dtype)[:, :2]
        mix = None
        if not a.no_mix and nsig * elt > 128 and G.tile_stats and G.tile_stats.get("enabled"):
            try:
                acc = {"real": [0.0, 0], 1: [0.0, 0], 2: [0.0, 0]}
                for rep in range(6):
                    for which in ("real", 1, 2):
                        if which == "real":
                            dev.cheby_filter_dev(c, bx.ptr, by.ptr, nsig, lmax)
                            t = ctx.last_timing()
                        else:
                            t = dev.bench_step_mix(c[0], bx.ptr, by.ptr, nsig, lmax, which)
                        if rep:
                            acc[which][0] += t["steps_ms"]
                            acc[which][1] += t["step_launches"]
                mix = {k_: v[0] / max(v[1], 1) for k_, v in acc.items()}
            except Exception as e:
                mix = {"error": repr(e)}
    finally:
        bx.free()
        by.free()
    U = N * nsig * elt
    b_launch = (dev.nnz_l * (elt + 4) + 4 * (N + 1) + 3 * U) + U / K
    avg = steps_ms / max(launches, 1)
    res = {"dtype": "f32" if elt == 4 else "f64", "value": N * nsig * K * a.steps / elapsed,
           "unit": "vertex*signal*order/s", "ms_per_step": elapsed / a.steps * 1e3, "steps": a.steps,
           "frac_whole_call": b_launch * K / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS,
           "roofline": {"bound": "hbm", "achieved": b_launch / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": b_launch / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": b_launch,
                        "avg_launch_ms": avg, "launches_timed": launches, "traffic": None},
           "note": "same graph, coefficients and signals as the headline, computed in this dtype (its own device "
                   "Laplacian and tiles); roofline from the HIP-event time of the recurrence launches"}
    res["auto_evaluation"] = auto
    if newton_ms is not None:  # (key kept from round 5: the evaluation 'auto' picked, Newton or product form)
        res["newton_form"] = {"evaluation": auto, "ms_per_order": newton_ms,
                              "frac_of_8TBps": b_launch / (newton_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "guard": (filters.newton_guard if auto == "newton" else filters.product_guard)(c[0], dtype)[1]}
    if mix is not None and "error" not in mix:
        res["roofline"].update(mix_launch_ms=mix[1], mix_nobarrier_launch_ms=mix[2], step_launch_ms_beside_mix=mix["real"],
                               frac_of_mix_ceiling=mix[1] / mix["real"],
                               mix_ceiling_frac=b_launch / (mix[1] * 1e-3) / 1e9 / HBM_PEAK_GBS)
    elif mix is not None:
        res["roofline"]["mix_error"] = mix["error"]
    if oracle:
        from oracle import cheby_oracle as orc
        ref = orc.cheby_op(G.L.astype(np.float64), lmax, c[0], x[:, :2].astype(np.float64))
        res["parity_vs_oracle"] = {"max_rel_err": float(np.max(np.abs(y - ref)) / np.max(np.abs(ref))), "columns": 2,
                                   "tolerance": 1e-3 if elt == 4 else 1e-5}
        if y_newton is not None:
            res["newton_form"]["parity_vs_oracle"] = {
                "max_rel_err": float(np.max(np.abs(y_newton - ref)) / np.max(np.abs(ref))), "columns": 2,
                "tolerance": 1e-3 if elt == 4 else 1e-5}
    for dt, g_ in list(G._dev.items()):  # the extra device graph goes; the headline's stays
        if dt == np.dtype(dtype) and dt != G.compute_dtype:
            g_.destroy()
            del G._dev[dt]
    return res


def chain3(a, G, x, K, oracle=True):
    """heat -> MexicanHat(Nf=4) analysis -> synthesis through Filter.filter (the chain of the doctest, filter.py:232-256)
    on the headline graph and signals: numpy arrays in and out of every call against ONE upload, three device-resident
    calls and ONE download (engine.DeviceArray)."""
    from pygsp_amd import filters
    heat, bank = filters.Heat(G, a.scale), filters.MexicanHat(G, Nf=4)
    N, nsig = x.shape

    def on_device():
        d = G.to_device(x)
        t1 = time.perf_counter()
        y = bank.synthesize(bank.analyze(heat.filter(d, order=K), order=K), order=K)
        G.context.sync()
        t2 = time.perf_counter()
        out = np.asarray(y)
        return out, t1, t2

    def on_host():
        return bank.synthesize(bank.analyze(heat.filter(x, order=K), order=K), order=K)

    on_device()  # warm-up (workspaces)
    best_dev = best_calls = None
    for _ in range(2):
        t0 = time.perf_counter()
        y_dev, t1, t2 = on_device()
        dt = time.perf_counter() - t0
        if best_dev is None or dt < best_dev:
            best_dev, best_calls = dt, t2 - t1
    on_host()
    t0 = time.perf_counter()
    y_host = on_host()
    t_host = time.perf_counter() - t0
    res = {"workload": "Heat({:g}) -> MexicanHat(Nf=4) analysis -> synthesis, order {}, {} x {} signals, {}".format(
               a.scale, K, N, nsig, a.dtype),
           "chain3_device_resident_ms": best_dev * 1e3, "of_which_three_filter_calls_ms": best_calls * 1e3,
           "chain3_host_arrays_ms": t_host * 1e3, "speedup": t_host / best_dev,
           "identical_bits": bool(np.array_equal(y_dev, y_host)),
           "note": "device-resident: G.to_device(x) (one upload), three Filter.filter calls on DeviceArrays, np.asarray "
                   "(one download); host arrays: the same three calls on numpy arrays (six PCIe crossings, the (N, Nsig, "
                   "4) analysis result materialised on the host)"}
    if oracle:
        from oracle import cheby_oracle as orc
        L, lm = G.L.astype(np.float64), float(G.lmax)
        kern = orc.mexican_hat_kernels(lm, 4)
        r = orc.filter_chebyshev(L, lm, [orc.heat_kernel(a.scale, lm)], x[:, :1].astype(np.float64), K)
        r = orc.filter_chebyshev(L, lm, kern, orc.filter_chebyshev(L, lm, kern, r, K), K)
        res["parity_vs_oracle"] = {"max_rel_err": float(np.max(np.abs(y_dev[:, 0] - r)) / np.max(np.abs(r))),
                                   "columns": 1, "tolerance": 1e-5 if a.dtype == "f64" else 1e-3}
    return res


class RankWork:
    """One rank's share of the headline workload, resident on one libgspx context: an independent sensor graph
    (device k-NN -> device Laplacian -> tiles), Heat coefficients, the input panel and the output buffer."""

    def __init__(self, a, ctx, rank, dtype, output=True):
        from pygsp_amd import engine, filters, graphs
        self.ctx, self.rank, self.dtype = ctx, rank, dtype
        N, nsig, K = a.n, a.nsig, a.order
        self.coords = np.random.default_rng(42 + rank).uniform(0, 1, (N, 2))  # nngraphs/sensor.py:56-70
        t0 = time.perf_counter()
        self.W, _, self.knn_info = engine.knn_graph(self.coords, a.knn, ctx=ctx)
        self.t_gen_dev = time.perf_counter() - t0
        t0 = time.perf_counter()
        self.G = graphs.Graph(self.W, coords=self.coords, compute_dtype=dtype, ctx=ctx, reorder=a.reorder,
                              tiles="auto" if a.tiles == "auto" else False)
        self.t_graph = time.perf_counter() - t0
        self.dev = self.G.device_graph()
        t0 = time.perf_counter()
        self.G.estimate_lmax("bounds")
        self.t_lmax_bounds = time.perf_counter() - t0
        self.lmax = float(self.G.lmax)
        # Heat(scale) coefficients, compute_cheby_coeff (approximations.py:9-55)
        self.c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(self.G, a.scale), m=K))
        self.x = np.random.default_rng(1234 + rank).standard_normal((N, nsig)).astype(dtype)  # zeros would clock higher
        self.bx = ctx.upload(self.x)
        self.by = ctx.alloc(self.x.nbytes) if output else None
        self.nsig = nsig
        self.dev_ms = self.steps_ms = 0.0
        self.launches = 0
        self.tuning = None

    def tune(self, candidates, y_ptr=None, stride_mb=0):
        """Set-up, outside every timed region: draw `candidates` physical backings for the context's streamed
        workspaces, run this rank's own call on each and keep the fastest (DeviceGraph.tune_placement;
        profiles/r06_placement.md).  No-op for the plain kernels' graphs and for candidates < 2."""
        tiled = bool(self.G.tile_stats and self.G.tile_stats.get("enabled"))
        # (panels within the 256 MB Infinity Cache do not feel where they lie in HBM: nothing to tune)
        if candidates < 2 or not tiled or (y_ptr is None and self.by is None) or self.x.nbytes < (192 << 20):
            return None
        t0 = time.perf_counter()
        try:
            rep = self.dev.tune_placement(self.c[0], self.bx.ptr, y_ptr or self.by.ptr, self.nsig, self.lmax, candidates,
                                          stride_mb)
        except Exception as e:  # a tuning step: never a reason to lose the measurement
            self.tuning = {"error": repr(e)}
            return self.tuning
        N, elt = self.x.shape[0], self.x.dtype.itemsize
        U = N * self.nsig * elt
        K = self.c.shape[1] - 1
        b_launch = self.dev.nnz_l * (elt + 4) + 4 * (N + 1) + 3 * U + U / K
        self.tuning = {"seconds": time.perf_counter() - t0, "candidates": candidates, "stride_mb": stride_mb,
                       "kept": rep["kept"], "candidates_launch_ms": rep["launch_ms"],
                       "candidates_frac": [(b_launch / (v * 1e-3) / 1e9 / HBM_PEAK_GBS) if v > 0 else None
                                           for v in rep["launch_ms"]],
                       "what": "DeviceGraph.tune_placement: physical backings of the streamed workspaces drawn in the "
                               "set-up - a pad of stride_mb held between two draws, so that they sample the card's memory "
                               "zones -, this very call run on each, the fastest kept, everything else released "
                               "(bit-identical results; candidate 0 = the first draw, what a run without tuning would "
                               "have used; null = not drawn, memory ran out)"}
        return self.tuning

    def step(self, y_ptr=None):
        return self.dev.cheby_filter_dev(self.c, self.bx.ptr, y_ptr or self.by.ptr, self.nsig, self.lmax)

    def step_timed(self):
        self.dev_ms += self.step()
        t = self.ctx.last_timing()
        self.steps_ms += t["steps_ms"]
        self.launches += t["step_launches"]

    def parity(self, cols=2):
        from oracle import cheby_oracle as orc
        ref = orc.cheby_op(self.G.L.astype(np.float64), self.lmax, self.c[0], self.x[:, :cols].astype(np.float64))
        y = self.by.download(self.x.shape, self.dtype)[:, :cols]
        return float(np.max(np.abs(y - ref)) / np.max(np.abs(ref)))

    def free(self):
        self.bx.free()
        if self.by is not None:
            self.by.free()
        for g_ in list(self.G._dev.values()):
            g_.destroy()
        self.G._dev = {}


def run_signal_parallel(group, a, dtype, no_parity=False):
    """SURVEY 8(e)(2), strong scaling of the HEADLINE call: ONE 1M-vertex graph replicated on every context of the
    group, its 64 signal columns split over them (device resident), one gather of the column blocks onto context 0
    (gspx_gather).  The same machinery as Filter.filter(..., devices=[...]) (pygsp_amd.multi), timed like the
    headline: thread barrier, K steps, device syncs, first start to last finish."""
    import threading

    from pygsp_amd import dist as gdist
    from pygsp_amd import engine, filters, graphs
    n = len(group)
    N, nsig, K = a.n, a.nsig, a.order
    coords = np.random.default_rng(42).uniform(0, 1, (N, 2))
    W, _, _ = engine.knn_graph(coords, a.knn, ctx=group.ctxs[0])
    G = graphs.Graph(W, coords=coords, compute_dtype=dtype, ctx=group.ctxs[0], reorder=a.reorder,
                     tiles="auto" if a.tiles == "auto" else False)
    G.estimate_lmax("bounds")
    lmax = float(G.lmax)
    c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, a.scale), m=K))
    x = np.random.default_rng(1234).standard_normal((N, nsig)).astype(dtype)
    cols = [gdist.shard_units(nsig, r, n) for r in range(n)]
    state = [None] * n

    def setup(i, ctx):
        if len(cols[i]) == 0:
            return
        dev = G.device_graph() if i == 0 else engine.DeviceGraph.from_w(G.W, G.lap_type, dtype=dtype,
                                                                       perm=G._internal_order(), ctx=ctx)
        if i > 0 and a.tiles == "auto":
            dev.auto_gather_tiles()
        xs = np.ascontiguousarray(x[:, cols[i].start:cols[i].stop])
        bx, by = ctx.upload(xs), ctx.alloc(xs.nbytes)
        state[i] = (dev, bx, by, xs.shape[1])
        for _ in range(max(a.warmup, 1)):
            dev.cheby_filter_dev(c, bx.ptr, by.ptr, xs.shape[1], lmax)
        ctx.sync()

    group.run(setup)
    bar = threading.Barrier(n)

    def timed(i, ctx):
        bar.wait()
        t0 = time.perf_counter()
        if state[i] is not None:
            dev, bx, by, w = state[i]
            for _ in range(a.steps):
                dev.cheby_filter_dev(c, bx.ptr, by.ptr, w, lmax)
            ctx.sync()
        return t0, time.perf_counter()

    spans = group.run(timed)
    elapsed = max(t1 for _, t1 in spans) - min(t0 for t0, _ in spans)
    root_buf, t_g, impl = group.gather([st[2] if st else None for st in state], 0)
    root_buf.free()
    root_buf, t_g, impl = group.gather([st[2] if st else None for st in state], 0)
    err = None
    if not no_parity:
        from oracle import cheby_oracle as orc
        flat = root_buf.download((N * nsig,), dtype)
        y = np.empty((N, nsig), dtype=dtype)
        off = 0
        for cr in cols:
            w = len(cr)
            y[:, cr.start:cr.stop] = flat[off:off + N * w].reshape(N, w)
            off += N * w
        check = sorted({0, nsig - 1})  # the first column of the first GPU and the last of the last
        ref = orc.cheby_op(G.L.astype(np.float64), lmax, c[0], x[:, check].astype(np.float64))
        err = float(np.max(np.abs(y[:, check] - ref)) / np.max(np.abs(ref)))
    root_buf.free()
    for i, st in enumerate(state):
        if st is not None:
            st[1].free()
            st[2].free()
            if i > 0:
                st[0].destroy()
    for g_ in list(G._dev.values()):
        g_.destroy()
    G._dev = {}
    return {"workload": "ONE Sensor(N={}, k={}) graph replicated on {} GPU(s), its {} signals split {} (strong scaling of "
                        "the headline call)".format(N, a.knn, n, nsig, "/".join(str(len(cr)) for cr in cols)),
            "n_gpus": n, "ms_per_step": elapsed / a.steps * 1e3, "value": N * nsig * K * a.steps / elapsed,
            "unit": "vertex*signal*order/s", "gather_ms": t_g * 1e3, "gather_impl": impl,
            "parity_vs_oracle": {"max_rel_err": err, "columns": 2, "tolerance": 1e-5 if a.dtype == "f64" else 1e-3}}


def run_batch_config_threads(group, no_parity=False):
    """BASELINE.json configs[4] from ONE process: the batch of 8 independent Sensor(N=500000) graphs x 32 signals,
    Heat order 30, sharded over the group's contexts (one driver thread each, no data-path collective), the 8
    output blocks gathered onto context 0 by gspx_gather.  Strong scaling of a fixed batch."""
    import threading

    from pygsp_amd import dist as gdist
    from pygsp_amd import filters, graphs
    n_graphs, N5, nsig, K = 8, 500000, 32, 30
    n = len(group)
    block = N5 * nsig * 8
    shards = [list(gdist.shard_units(n_graphs, r, n)) for r in range(n)]
    state = [None] * n

    def setup(i, ctx):
        by_all = ctx.alloc(max(len(shards[i]), 1) * block)
        jobs = []
        for slot, g in enumerate(shards[i]):
            G = graphs.Sensor(N5, seed=g, compute_dtype=np.float64, ctx=ctx)
            G.estimate_lmax("bounds")
            c = np.atleast_2d(filters.compute_cheby_coeff(filters.Heat(G, 50), m=K))
            x = np.random.default_rng(100 + g).standard_normal((N5, nsig))
            jobs.append((G, c, x, ctx.upload(x), by_all.ptr + slot * block))
        state[i] = (by_all, jobs)
        run_all(i)
        ctx.sync()

    def run_all(i):
        for G, c, _, bx, y_ptr in state[i][1]:
            G.device_graph().cheby_filter_dev(c, bx.ptr, y_ptr, nsig, float(G.lmax))

    group.run(setup)
    reps = 3
    bar = threading.Barrier(n)

    def timed(i, ctx):
        bar.wait()
        t0 = time.perf_counter()
        for _ in range(reps):
            run_all(i)
        ctx.sync()
        return t0, time.perf_counter()

    spans = group.run(timed)
    wall = (max(t1 for _, t1 in spans) - min(t0 for t0, _ in spans)) / reps
    # the gather hands whole buffers over: a context with fewer graphs than slots would ship nothing extra
    root_buf, t_g, impl = group.gather([st[0] if shards[i] else None for i, st in enumerate(state)], 0)
    root_buf.free()
    root_buf, t_g, impl = group.gather([st[0] if shards[i] else None for i, st in enumerate(state)], 0)
    err = None
    if not no_parity:
        from oracle import cheby_oracle as orc
        got = root_buf.download((n_graphs, N5, nsig), np.float64)
        err = 0.0
        for i in range(n):  # the first graph of every context, one column, as it arrived on the root
            if not shards[i]:
                continue
            G, c, x, _, _ = state[i][1][0]
            ref = orc.cheby_op(orc.laplacian(G.W), float(G.lmax), c[0], x[:, :1])
            y = got[shards[i][0]]
            err = max(err, float(np.max(np.abs(y[:, :1] - ref.reshape(N5, 1))) / np.max(np.abs(ref))))
    root_buf.free()
    for by_all, jobs in state:
        for G, _, _, bx, _ in jobs:
            bx.free()
            for g_ in list(G._dev.values()):
                g_.destroy()
            G._dev = {}
        by_all.free()
    return {"workload": "configs[4]: 8 x Sensor(N=500000), Heat(50) order 30, 32 signals each, f64, sharded {} per GPU "
                        "(strong scaling of a fixed batch)".format("/".join(str(len(s_)) for s_ in shards)),
            "n_graphs": n_graphs, "n_gpus": n, "ms": wall * 1e3, "value": n_graphs * N5 * nsig * K / wall,
            "unit": "vertex*signal*order/s", "gather_ms": t_g * 1e3, "gather_impl": impl,
            "parity_vs_oracle": {"max_rel_err": err, "columns": 1, "graphs_checked": sum(1 for s_ in shards if s_),
                                 "tolerance": 1e-5}}


def main_threads(a):
    """`python bench.py --gpus N` without a launcher: ONE process, one libgspx context and one driver thread per
    GPU (ctypes releases the GIL inside the library), the final gather through gspx_gather - RCCL
    (ncclCommInitAll, grouped send / recv) inside libgspx.  No torch on this path.  Refuses to run on fewer GPUs
    than asked for."""
    import threading

    from pygsp_amd import _capi, multi
    devices = [int(d) for d in a.devices.split(",")] if a.devices else list(range(a.gpus))
    if len(devices) != a.gpus:
        raise SystemExit("bench.py: --gpus {} but --devices names {} entries".format(a.gpus, len(devices)))
    visible = _capi.device_count()
    if visible < 1 or max(devices) >= visible or min(devices) < 0:
        raise SystemExit("bench.py: --gpus {} needs HIP devices {} but only {} visible - refusing to report a "
                         "{}-GPU number from fewer GPUs".format(a.gpus, sorted(set(devices)), visible, a.gpus))
    group = multi.DeviceGroup(devices)
    n = len(group)
    dtype = np.float64 if a.dtype == "f64" else np.float32
    elt = np.dtype(dtype).itemsize
    N, nsig, K = a.n, a.nsig, a.order
    ranks = [None] * n

    def setup(i, ctx):
        for kv in a.opt:
            k, v = kv.split("=")
            ctx.set_option(k, int(v))
        ranks[i] = RankWork(a, ctx, i, dtype)
        ranks[i].tune(a.tune_candidates, None, a.tune_stride_mb)
        for _ in range(a.warmup):
            ranks[i].step()
        ctx.sync()

    group.run(setup)
    bar = threading.Barrier(n)

    def timed(i, ctx):
        r = ranks[i]
        bar.wait()  # every context idle, every thread here: the timed region starts
        t0 = time.perf_counter()
        for _ in range(a.steps):
            r.step_timed()
        ctx.sync()
        return t0, time.perf_counter()

    spans = group.run(timed)
    elapsed = max(t1 for _, t1 in spans) - min(t0 for t0, _ in spans)  # first start to last finish

    # ---- the path's one collective, outside the timed region: every GPU's output block -> context 0 ----
    gather = None
    if not a.no_gather:
        root_buf, t_first, impl = group.gather([r.by for r in ranks], 0)  # first call builds the communicators
        root_buf.free()
        root_buf, t_g, impl = group.gather([r.by for r in ranks], 0)
        got = root_buf.download((n, N, nsig), dtype)
        for i in (0, n - 1):  # as it arrived: the root's own block and the last peer's, bit for bit
            assert np.array_equal(got[i], ranks[i].by.download((N, nsig), dtype)), "gathered block {} differs".format(i)
        del got
        root_buf.free()
        uses_rccl = "RCCL" in impl
        gather = {"gather_ms": t_g * 1e3, "gather_first_call_ms": t_first * 1e3, "gather_impl": impl,
                  "rccl_ranks": group.n_distinct if uses_rccl else 0,
                  "gather_GBps": (n - 1) * N * nsig * elt / max(t_g, 1e-9) / 1e9 if group.n_distinct > 1 else None}
    parity = None
    if not a.no_cpu:
        errs = [r.parity(2) for r in ranks]
        parity = {"max_rel_err": max(errs), "columns": 2, "ranks": n, "tolerance": 1e-5 if a.dtype == "f64" else 1e-3}

    # ---- per-device calibration, all GPUs at once like the timed region: copy rate, and the step beside its mix
    # ceiling (gspx_bench_step_mix: the same launches with the row products removed), real and calibration calls
    # alternating
    def calibrate(i, ctx):
        r = ranks[i]
        res = {"copy_GBps": None, "mix_launch_ms": None, "frac_of_mix_ceiling": None}
        try:
            res["copy_GBps"] = ctx.bench_copy(1 << 30, 5)
            tiled_i = bool(r.G.tile_stats and r.G.tile_stats.get("enabled"))
            if not a.no_mix and tiled_i and nsig * elt > 128 and (nsig * elt) % 16 == 0:
                acc = {"real": [0.0, 0], 1: [0.0, 0]}
                for rep in range(5):
                    for which in ("real", 1):
                        if which == "real":
                            r.step()
                            t = ctx.last_timing()
                        else:
                            t = r.dev.bench_step_mix(r.c[0], r.bx.ptr, r.by.ptr, nsig, r.lmax, 1)
                        if rep:
                            acc[which][0] += t["steps_ms"]
                            acc[which][1] += t["step_launches"]
                real_ms, mix_ms = (acc[k_][0] / max(acc[k_][1], 1) for k_ in ("real", 1))
                res.update(mix_launch_ms=mix_ms, step_launch_ms_beside_mix=real_ms, frac_of_mix_ceiling=mix_ms / real_ms)
                r.step()  # the real result back in the output buffer
                ctx.sync()
        except Exception as e:
            res["error"] = repr(e)
        return res

    calib = group.run(calibrate)
    # ---- HBM traffic of the step (rocprofv3 child passes on the first GPU) and the CPU baseline, as at N = 1 ----
    traffic = traffic_source = None
    default_workload = (N, nsig, K, a.knn, a.tiles) == (1000000, 64, 30, 8, "auto")
    if default_workload and not a.no_live_traffic:
        group.ctxs[0].sync()
        live = live_traffic(a.dtype, device=devices[0])
        if live is not None:
            traffic = live[0]
            traffic_source = {"how": "measured in this run on GPU {}: the N = 1 call under rocprofv3 --pmc FETCH_SIZE and "
                                     "--pmc WRITE_SIZE (two child runs), (2*FETCH_SIZE + WRITE_SIZE) per k_step_tile "
                                     "launch".format(devices[0]), "launches": live[1], "calibration": live[2]}
    cpu_baseline = None
    if not a.no_cpu:
        from oracle import cheby_oracle as orc
        r0_ = ranks[0]
        cols = min(a.cpu_cols, nsig)
        Lh = r0_.G.L.astype(np.float64)
        xs = r0_.x[:, :cols].astype(np.float64)
        orc.cheby_op(Lh, r0_.lmax, r0_.c[0], xs[:, :1])
        tc = time.perf_counter()
        ref = orc.cheby_op(Lh, r0_.lmax, r0_.c[0], xs)
        t_cpu = time.perf_counter() - tc
        y0 = r0_.by.download(r0_.x.shape, dtype)[:, :cols]
        cpu_baseline = {"value": N * cols * K / t_cpu, "unit": "vertex*signal*order/s", "cores": 1, "kind": "port",
                        "sample": "oracle port of the reference (scipy csr_matvecs, single-threaded like the reference): "
                                  "the first GPU's graph / coefficients, first {} of {} signal columns, order {}, float64 "
                                  "({} host cores present), {:.1f} s".format(cols, nsig, K, os.cpu_count(), t_cpu),
                        "note": CPU_BASELINE_NOTE,
                        "parity_of_the_sample": float(np.max(np.abs(y0 - ref)) / np.max(np.abs(ref)))}
        del Lh, xs, ref, y0

    from pygsp_amd import engine
    seen = engine.comm_info()  # after the gather: the device set ncclCommInitAll built, as RCCL reports it
    r0 = ranks[0]
    nnz_l = r0.dev.nnz_l
    U = N * nsig * elt
    b_alg_launch = (K * (nnz_l * (elt + 4) + 4 * (N + 1) + 3 * U) + U) / K
    launches = sum(r.launches for r in ranks)
    avg_launch_ms = sum(r.steps_ms for r in ranks) / max(launches, 1)
    achieved = b_alg_launch / (avg_launch_ms * 1e-3) / 1e9
    tiled = bool(r0.G.tile_stats and r0.G.tile_stats.get("enabled"))
    out = {
        "metric": baseline_metric(), "value": n * N * nsig * K * a.steps / elapsed, "unit": "vertex*signal*order/s",
        "n_gpus": n, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
        "config": {
            "workload": "Sensor(N={}, k={}) combinatorial Laplacian, Heat(scale={:g}) order {}, {} signals, "
                        "device-resident (north-star headline), one independent graph per GPU".format(N, a.knn, a.scale, K, nsig),
            "N": N, "Nsig": nsig, "order": K, "Nf": 1, "nnz_W": int(r0.W.nnz), "nnz_L": int(nnz_l),
            "n_edges": int(r0.G.n_edges), "lmax": r0.lmax, "lmax_method": "bounds",
            "parallelism": "graph-parallel x{} (independent graphs, no data-path collective; one final gather)".format(n),
            "evaluation": "recurrence", "engine_options": a.opt, "gather_tiles": r0.G.tile_stats,
            "gather_impl": gather["gather_impl"] if gather else None, "rccl_version": seen["rccl_version"],
            "rccl_nranks_seen": seen["nranks"], "distinct_devices": group.n_distinct,
            "real_pygsp_on_device": REAL_PYGSP_RECORD},
        "launcher": "one process, one driver thread + one libgspx context per GPU (no torch)",
        "devices": devices,
        "driver_thread_cores": [len(p) if p else None for p in group.pinned],  # NUMA pinning (multi.pin_thread_near)
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "traffic_over_algorithmic": (traffic / b_alg_launch) if traffic else None,
                     "kernel": "k_step_tile" if tiled else "k_step_panel / k_step_lds",
                     "algorithmic_bytes_per_launch": b_alg_launch, "avg_launch_ms": avg_launch_ms,
                     "launches_timed": launches, "note": "average over the launches of all GPUs",
                     # the same bytes over the whole job's wall time per step (launch gaps, host, the slowest GPU)
                     "frac_whole_call": n * K * b_alg_launch / (elapsed / a.steps) / 1e9 / (n * HBM_PEAK_GBS),
                     "copy_GBps_this_run": calib[0].get("copy_GBps"),
                     "copy_GBps_per_device": [c_.get("copy_GBps") for c_ in calib],
                     "frac_of_mix_ceiling": calib[0].get("frac_of_mix_ceiling"),
                     "frac_of_mix_ceiling_per_device": [c_.get("frac_of_mix_ceiling") for c_ in calib],
                     "mix_launch_ms_per_device": [c_.get("mix_launch_ms") for c_ in calib],
                     "parity_max_rel_err": parity["max_rel_err"] if parity else None,
                     "parity_tolerance": parity["tolerance"] if parity else None},
        "per_device": [{"device": devices[i], "ms_per_step": (spans[i][1] - spans[i][0]) / a.steps * 1e3,
                        "device_ms_per_step": r.dev_ms / a.steps, "avg_launch_ms": r.steps_ms / max(r.launches, 1),
                        "frac": b_alg_launch / (r.steps_ms / max(r.launches, 1) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        **calib[i]}
                       for i, r in enumerate(ranks)],
        "parity_vs_oracle": parity,
        "cpu_baseline": cpu_baseline,
        "setup_s": {"placement_tuning": [r.tuning for r in ranks]},
    }
    if gather:
        out.update(gather)
    else:
        out.update(gather_ms=None, gather_impl=None, rccl_ranks=0)
    for r in ranks:
        r.free()
    if not a.no_configs:
        try:
            out["signal_parallel"] = run_signal_parallel(group, a, dtype, no_parity=a.no_cpu)
        except Exception as e:  # an extra: never a reason to lose the headline measurement
            out["signal_parallel"] = {"error": repr(e)}
        try:
            out["batch_config4"] = run_batch_config_threads(group, no_parity=a.no_cpu)
        except Exception as e:
            out["batch_config4"] = {"error": repr(e)}
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)  # RCCL's banner (printf) before the result line
    except Exception:
        pass
    print(json.dumps(out), flush=True)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit("--gpus {} but WORLD_SIZE={}".format(a.gpus, world))
    if world == 1 and a.gpus > 1:  # no launcher: this process drives the N GPUs itself
        return main_threads(a)

    # all host cores first: its workers are forked, so it runs before this process touches HIP
    cpu_all = None
    if rank == 0 and world == 1 and not a.no_cpu and a.cpu_all_cores != 0 and not a.no_headline:
        # (one process per core up to 64: the port is memory-bound, 256 processes on the 256-core host of
        # the GPU box delivered 3.4x one core and took 50 s; the sample is sized to stay within ~10-15 s)
        procs = a.cpu_all_cores if a.cpu_all_cores > 0 else min(os.cpu_count() or 1, 64)
        try:
            cpu_all = cpu_all_cores(a.n, a.knn, a.order, a.scale, procs)
        except Exception as e:  # a reported baseline, never a reason to lose the measurement
            cpu_all = {"error": repr(e)}

    from pygsp_amd import engine, graphs
    from tools import torchrun_plumbing as gdist  # launcher plumbing (torch only when WORLD_SIZE > 1)

    if a.no_headline:  # profiling passes of single configs
        ctx0 = engine.default_context(local)
        for kv in a.opt:
            k, v = kv.split("=")
            ctx0.set_option(k, int(v))
        if a.calibrate_copy:
            ctx0.bench_copy(1 << 29, 2)
        print(json.dumps({"configs": run_configs(ctx0, a.only_config, a.config_reps, a.config_oracle_cols)}))
        return

    torch = None
    tdev = None
    if world > 1:
        import torch  # plumbing only: rendezvous, barrier, MAX-reduce, RCCL gather
        if os.environ.get("GSPX_ALL_RANKS_DEVICE0"):  # test hook: several ranks on one GPU (gloo)
            local = 0
        gdist.init_process_group(a.backend)
        tdev = torch.device("cuda", local)
    rdev = tdev if a.backend == "nccl" else None  # where the scalar reductions live

    dtype = np.float64 if a.dtype == "f64" else np.float32
    elt = np.dtype(dtype).itemsize
    N, nsig, K = a.n, a.nsig, a.order

    # ---- synthetic workload: one independent sensor graph per rank -----------------------------
    from pygsp_amd import _capi
    if local >= _capi.device_count():
        raise SystemExit("bench.py: rank {} wants HIP device {} but only {} visible - refusing to run".format(
            rank, local, _capi.device_count()))
    ctx = engine.default_context(local)
    for kv in a.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    # the graph's weights come from the device k-NN path ...
    rw = RankWork(a, ctx, rank, dtype, output=(world == 1))
    coords, W, knn_info, t_gen_dev, t_graph = rw.coords, rw.W, rw.knn_info, rw.t_gen_dev, rw.t_graph
    G, dev, lmax, c, x, bx, by = rw.G, rw.dev, rw.lmax, rw.c, rw.x, rw.bx, rw.by
    t_gen, knn_diff = None, None
    if world == 1:  # ... and are checked against the host (KD-tree) construction of the same matrix
        t0 = time.perf_counter()
        Wh, ch = graphs.sensor_weights(N, k=a.knn, seed=42 + rank)
        t_gen = time.perf_counter() - t0
        assert np.array_equal(ch, coords)
        knn_diff = float(abs(Wh - W).max()) if Wh.nnz == W.nnz else float("inf")
        del Wh
    t_lanczos = lanczos_ratio = None
    from pygsp_amd import filters
    if torch is not None:
        ty = torch.empty((1, N, nsig), dtype=torch.float64 if a.dtype == "f64" else torch.float32,
                         device=tdev)
        y_ptr = ty.data_ptr()
    else:
        y_ptr = by.ptr

    nodes, dcoef = filters.cheb_to_newton(c[0])

    def step_recurrence():
        return dev.cheby_filter_dev(c, bx.ptr, y_ptr, nsig, lmax)

    def step_newton():
        return dev.newton_filter_dev(nodes, dcoef, bx.ptr, y_ptr, nsig, lmax)

    product_ok, product_guard = filters.product_guard(c[0], dtype)
    program = filters.cheb_to_product(c[0], dtype) if product_guard.get("finite") else None

    def step_product():
        return dev.program_filter_dev(program, bx.ptr, y_ptr, nsig, lmax)

    step = step_newton if a.evaluation == "newton" else step_recurrence

    def download_y():  # this rank's output block as a host array (N, nsig)
        if torch is not None:
            torch.cuda.synchronize(tdev)
            return ty.cpu().numpy()[0]
        return by.download((1, N, nsig), dtype)[0]

    def fence():
        ctx.sync()
        if torch is not None:
            torch.cuda.synchronize(tdev)
            gdist.barrier()

    if a.calibrate_copy:  # k_permute_in as a pure copy of 2 x 512 MiB: known bytes for the PMC passes
        ctx.bench_copy(1 << 29, 2)
    if a.evaluation == "recurrence":
        rw.tune(a.tune_candidates, y_ptr, a.tune_stride_mb)  # set-up: the workspaces' backing (setup_s.placement_tuning)
    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    dev_ms, steps_ms, launches = 0.0, 0.0, 0
    for _ in range(a.steps):
        dev_ms += step()
        t = ctx.last_timing()
        steps_ms += t["steps_ms"]
        launches += t["step_launches"]
    fence()
    elapsed = time.perf_counter() - t0
    if torch is not None:
        elapsed = gdist.max_over_ranks(elapsed, rdev)
        steps_ms_max = gdist.max_over_ranks(steps_ms, rdev)
    else:
        steps_ms_max = steps_ms

    # ---- same polynomial in Newton form (extra, reported separately; not the headline) ----------
    def time_form(step_form):
        for _ in range(a.warmup):
            step_form()
        fence()
        tn = time.perf_counter()
        n_ms = 0.0
        for _ in range(a.steps):
            step_form()
            n_ms += ctx.last_timing()["steps_ms"]
        fence()
        n_elapsed = time.perf_counter() - tn
        if torch is not None:
            n_elapsed = gdist.max_over_ranks(n_elapsed, rdev)
        return n_elapsed, n_ms

    newton = y_newton = product = y_product = None
    if a.evaluation == "recurrence" and not a.no_newton:
        newton = time_form(step_newton)
        if rank == 0 and not a.no_cpu:  # its result, for the parity leg below (the recurrence overwrites y next)
            y_newton = download_y()[:, :min(a.cpu_cols, nsig)].copy()
        if program is not None:  # ... and as the product of its factors over its roots (filters.cheb_to_product)
            product = time_form(step_product)
            if rank == 0 and not a.no_cpu:
                y_product = download_y()[:, :min(a.cpu_cols, nsig)].copy()
    # ---- the mix ceiling of the recurrence step on THIS box (VERDICT r5 "Next 1"): the same call with the row
    # products removed from every launch (gspx_bench_step_mix: same grid, LDS-DMA tile loads, T_{k-2} / accumulator
    # loads, entry stream, stores, flushes and cache bits), mode 1 with the pass barriers, mode 2 without; real calls
    # alternate with the calibration calls so that all three see the same minutes of the same box
    mix = None
    tiled_now = bool(G.tile_stats and G.tile_stats.get("enabled"))
    if (a.evaluation == "recurrence" and not a.no_mix and tiled_now and nsig * elt > 128
            and (nsig * elt) % 16 == 0):
        try:
            import threading
            smi = {}
            th = threading.Thread(target=lambda: smi.update(smi_sample(local)), daemon=True)
            th.start()  # rocm-smi reads its sensors while the loop below keeps the GPU at the recurrence
            acc = {"real": [0.0, 0], 1: [0.0, 0], 2: [0.0, 0]}
            rep, t_mix0 = 0, time.perf_counter()
            while rep < 1 + max(3, min(a.steps, 10)) or (th.is_alive() and time.perf_counter() - t_mix0 < 5.0):
                for which in ("real", 1, 2):
                    if which == "real":
                        step_recurrence()
                        t = ctx.last_timing()
                    else:
                        t = dev.bench_step_mix(c[0], bx.ptr, y_ptr, nsig, lmax, which)
                    if rep:  # rep 0: warm-up
                        acc[which][0] += t["steps_ms"]
                        acc[which][1] += t["step_launches"]
                rep += 1
            th.join(10.0)
            mix = {k_: v[0] / max(v[1], 1) for k_, v in acc.items()}
            mix["smi"] = dict(smi)
            mix["read_GBps"] = ctx.bench_read(1 << 30, 5)
            # plain stream mixes, 256 MB per stream (gspx_bench_streams): reads : writes 1:0, 1:1, 3:1 - like the read-only
            # and copy rates they are the same in fast and slow memory zones (profiles/r06_placement.md): the box's
            # plain bandwidth, beside which the step's own figure is read
            mix["streams"] = {"r{}w{}".format(nr, nw): ctx.bench_streams(256 << 20, nr, nw, 0, 8, 3)
                              for nr, nw in ((1, 0), (1, 1), (3, 1))}
        except Exception as e:  # a calibration: never a reason to lose the measurement
            mix = {"error": repr(e)}
    if newton is not None or product is not None or mix is not None:
        step_recurrence()  # leave the headline result in y for the parity check / the gather below
        fence()

    # ---- the path's one collective, outside the timed region: outputs -> rank 0 ----------------
    gather_ms, gather_impl = None, None
    comm = None
    comm_seen = None
    if torch is not None and not a.no_gather:
        fence()
        # in the library: RCCL grouped send / recv (gspx_comm_gather); torch only carried the 128-byte id
        lib_mode = os.environ.get("GSPX_BENCH_LIB_GATHER", "1")  # "0": never, "force": also under gloo (tests)
        if (a.backend == "nccl" and lib_mode != "0") or lib_mode == "force":
            # run under a watchdog: a communicator that never forms (or a send that never completes) must not
            # cost the measurement that is already taken - after 240 s rank 0 prints what it has and every
            # rank leaves
            import threading
            box = {}
            uid = gdist.exchange_comm_id()  # the launcher's part; everything in the thread below is libgspx
            fence()

            def lib_gather():
                try:
                    box["comm"] = gdist.make_comm(ctx, uid)
                    root_buf = ctx.alloc(world * x.nbytes) if rank == 0 else None
                    ctx.sync()
                    tg = time.perf_counter()
                    box["comm"].gather(y_ptr, [x.nbytes] * world, 0, root_buf.ptr if rank == 0 else None)
                    box["ms"] = (time.perf_counter() - tg) * 1e3
                    if rank == 0:  # the root's own block, as it arrived through RCCL
                        got = root_buf.download((world, N, nsig), dtype)[0]
                        assert np.array_equal(got, ty.cpu().numpy()[0])
                        root_buf.free()
                    box["ok"] = True
                except Exception as e:  # agreed on below: every rank falls back together
                    sys.stderr.write("rank {}: in-library RCCL gather unavailable ({!r})\n".format(rank, e))
                # the agreement runs under the same watchdog: a rank that failed fast must not wait in a torch
                # collective for a peer that hangs inside ncclCommInitRank
                box["agree"] = gdist.sum_over_ranks(1.0 if box.get("ok") else 0.0, rdev)

            def no_id():
                box["agree"] = gdist.sum_over_ranks(0.0, rdev)

            th = threading.Thread(target=lib_gather if uid is not None else no_id, daemon=True)
            th.start()
            th.join(240.0)
            if th.is_alive():
                if rank == 0:
                    units = world * N * nsig * K * a.steps
                    avg = steps_ms / max(launches, 1)
                    b_alg = (dev.nnz_l * (elt + 4) + 4 * (N + 1) + 3 * N * nsig * elt) + N * nsig * elt / K
                    print(json.dumps({
                        "metric": baseline_metric(), "value": units / elapsed, "unit": "vertex*signal*order/s",
                        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": elapsed / a.steps * 1e3,
                        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": a.dtype,
                        "data": "synthetic",
                        "config": {"workload": "Sensor(N={}, k={}) Heat order {}, {} signals, device-resident (north-star "
                                               "headline)".format(N, a.knn, K, nsig)},
                        "roofline": {"bound": "hbm", "achieved": b_alg / (avg * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                     "unit": "GB/s", "frac": b_alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None},
                        "gather_ms": None, "gather_impl": "in-library RCCL gather did not finish within 240 s: "
                                                         "reported without the gather and the extras"}), flush=True)
                os._exit(0)
            comm = box.get("comm")
            if box.get("ok"):
                t_lib = box["ms"]
            if box.get("agree") != float(world) and comm is not None:
                comm.close()
                comm = None
        if comm is not None:
            gather_ms = gdist.max_over_ranks(t_lib, rdev)
            gather_impl = "libgspx gspx_comm_gather: RCCL grouped ncclSend/ncclRecv, one xGMI link per peer"
            comm_seen = comm.info()  # what RCCL itself says this communicator spans (ncclCommCount)
        else:
            fence()
            tg = time.perf_counter()
            blocks = gdist.gather_to_root(ty if a.backend == "nccl" else ty.cpu(), dst=0)
            torch.cuda.synchronize(tdev)
            gather_ms = gdist.max_over_ranks((time.perf_counter() - tg) * 1e3, rdev)
            if rank == 0:
                assert len(blocks) == world
            del blocks
            gather_impl = "torch.distributed ({}) send/recv".format("RCCL" if a.backend == "nccl" else a.backend)

    # ---- BASELINE configs[4]: a batch of 8 independent graphs sharded over the ranks ------------------
    batch5 = None
    if not a.no_configs and not a.no_cpu:
        batch5 = run_batch_config(local, rank, world, ctx, gdist, rdev, fence, comm)
    if comm is not None:
        comm.close()

    # N > 1: every rank checks two columns of its own output against the oracle (outside the timed region)
    parity_multi = None
    if torch is not None and not a.no_cpu:
        from oracle import cheby_oracle as orc
        ref = orc.cheby_op(G.L.astype(np.float64), lmax, c[0], x[:, :2].astype(np.float64))
        y2 = ty.cpu().numpy()[0][:, :2]
        err = float(np.max(np.abs(y2 - ref)) / np.max(np.abs(ref)))
        parity_multi = {"max_rel_err": gdist.max_over_ranks(err, rdev), "columns": 2, "ranks": world,
                        "tolerance": 1e-5 if a.dtype == "f64" else 1e-3}

    tiled = bool(G.tile_stats and G.tile_stats.get("enabled"))

    def form_report(r, form):
        ms_order = r[1] / (K * a.steps)
        guard_ok, guard = (filters.newton_guard(c[0], dtype) if form == "newton" else (product_ok, product_guard))
        out = {"note": ("same interpolating polynomial in Newton form (two-term Horner recurrence, no accumulator): "
                        "evaluation='newton', what evaluation='auto' runs for this call when the product form is refused "
                        "and filters.newton_guard clears the polynomial; parity_vs_oracle below is of THIS run")
               if form == "newton" else
                       ("same polynomial as the product of its factors over its roots (filters.cheb_to_product: a real "
                        "root is one step - gather h, write h': two panel passes -, a conjugate pair two steps; "
                        "guard.panel_passes_per_order against 3 2/3 of the recurrence): evaluation='product', what "
                        "evaluation='auto' runs for this call when filters.product_guard clears the polynomial; "
                        "ms_per_order is per order of the polynomial (K), not per launch"),
               "guard_ok": guard_ok, "guard": guard,
               "auto_picks": filters.choose_evaluation("auto", np.atleast_2d(c[0]), dtype, N, nsig),
               "value": world * N * nsig * K * a.steps / r[0], "ms_per_step": r[0] / a.steps * 1e3,
               "ms_per_order": ms_order,
               "achieved_GBps_alg": b_alg_launch / (ms_order * 1e-3) / 1e9,
               "frac_of_8TBps": b_alg_launch / (ms_order * 1e-3) / 1e9 / HBM_PEAK_GBS}
        return out

    # the streaming-copy rate of THIS box and process, right after the timed region: the recurrence runs at
    # a fixed fraction of it, and it moves by several percent between boxes
    copy_now = None
    try:
        copy_now = ctx.bench_copy(1 << 30, 5)
    except Exception:
        pass
    # N > 1: the slowest and the fastest GPU of the job, for the copy rate, the step and its mix ceiling (every rank
    # measures its own; rank 0's own values are the plain keys)
    spread = None
    if torch is not None:
        def lo_hi(v):
            v = float(v or 0.0)
            return [-gdist.max_over_ranks(-v, rdev), gdist.max_over_ranks(v, rdev)]
        mix_ok = mix is not None and "error" not in mix
        spread = {"copy_GBps_min_max_over_ranks": lo_hi(copy_now),
                  "avg_launch_ms_min_max_over_ranks": lo_hi(steps_ms / max(launches, 1)),
                  "frac_of_mix_ceiling_min_max_over_ranks": lo_hi(mix[1] / mix["real"] if mix_ok else 0.0),
                  "mix_launch_ms_min_max_over_ranks": lo_hi(mix[1] if mix_ok else 0.0)}

    # ---- roofline of the dominant kernel (the recurrence step) ---------------------------------
    nnz_l = dev.nnz_l
    U = N * nsig * elt
    csr = nnz_l * (elt + 4) + 4 * (N + 1)
    b_alg_call = K * (csr + 3 * U) + 1 * U
    b_alg_launch = b_alg_call / K
    avg_launch_ms = steps_ms / max(launches, 1)
    achieved = b_alg_launch / (avg_launch_ms * 1e-3) / 1e9
    traffic = None
    tfile = os.path.join(ROOT, "profiles", "traffic_{}.json".format(a.dtype))
    # the committed PMC measurement is of the default workload only
    default_workload = (N, nsig, K, a.knn, a.evaluation, a.tiles) == (1000000, 64, 30, 8, "recurrence", "auto")
    traffic_source = None
    live = None
    if default_workload and rank == 0 and not a.no_live_traffic:
        # (N > 1: the child pass runs on this rank's GPU while the other ranks wait at the next collective - the
        # timed region and the gather are over)
        ctx.sync()
        live = live_traffic(a.dtype, device=local)
    if live is not None:
        traffic = live[0]
        traffic_source = {"how": "measured in this run: the same call under rocprofv3 --pmc FETCH_SIZE and --pmc "
                                 "WRITE_SIZE (two separate child runs), (2*FETCH_SIZE + WRITE_SIZE) per k_step_tile launch",
                          "launches": live[1], "calibration": live[2]}
    elif default_workload and os.path.exists(tfile):
        try:
            traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
            traffic_source = ("profiles/traffic_{}.json: (2*FETCH_SIZE + WRITE_SIZE) per k_step_tile launch from "
                              "separate rocprofv3 --pmc passes of this same command (tools/gpu_prof.sh); PMC counters "
                              "cannot be read from inside this process").format(a.dtype)
        except Exception:
            traffic = None

    out = None
    if rank == 0:
        units = world * N * nsig * K * a.steps
        out = {
            "metric": baseline_metric(),
            "value": units / elapsed,
            "unit": "vertex*signal*order/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": a.dtype,
            "data": "synthetic",
            "config": {
                "workload": "Sensor(N={}, k={}) combinatorial Laplacian, Heat(scale={:g}) order {}, "
                            "{} signals, device-resident (north-star headline)".format(
                                N, a.knn, a.scale, K, nsig),
                "N": N, "Nsig": nsig, "order": K, "Nf": 1, "nnz_W": int(W.nnz), "nnz_L": int(nnz_l),
                "n_edges": int(G.n_edges), "lmax": lmax, "lmax_method": "bounds",
                "parallelism": "graph-parallel x{} (independent graphs, no data-path collective)".format(world),
                "internal_order": ("hilbert (2-D coordinates)" if coords.shape[1] == 2 else "morton")
                                  if G._internal_order() is not None else "none",
                "evaluation": a.evaluation,
                "engine_options": a.opt, "gather_tiles": G.tile_stats,
                # the path's one collective as this run executed it (self-validating on a multi-GPU node: the ranks RCCL
                # itself reports must equal n_gpus)
                "gather_impl": gather_impl, "rccl_version": (comm_seen or engine.comm_info())["rccl_version"],
                "rccl_nranks_seen": comm_seen["nranks"] if comm_seen else 0, "launcher_world_size": world,
                "real_pygsp_on_device": REAL_PYGSP_RECORD,
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_ceiling": achieved / HBM_COPY_GBS,
                "copy_GBps_this_run": copy_now,
                "frac_of_copy_this_run": (achieved / copy_now) if copy_now else None,
                "traffic": traffic, "traffic_source": traffic_source,
                "kernel": ("k_step_tile (one recurrence order per launch, gathered panel staged in LDS)"
                           if tiled else "k_step_panel / k_step_lds (one recurrence order per launch)"),
                "algorithmic_bytes_per_launch": b_alg_launch,
                "avg_launch_ms": avg_launch_ms, "launches_timed": launches,
            },
            "newton_form": None if newton is None else form_report(newton, "newton"),
            "product_form": None if product is None else form_report(product, "product"),
            "device_ms_per_step": dev_ms / a.steps,
            "device_ms_recurrence_per_step": steps_ms_max / a.steps,
            "gather_ms": gather_ms, "gather_impl": gather_impl,
            "setup_s": {"graph_generation_host": t_gen, "graph_generation_device_knn": t_gen_dev,
                        "device_knn_build_ms": knn_info["build_ms"], "device_knn_max_abs_diff_vs_host": knn_diff,
                        "graph_object_incl_device_laplacian": t_graph,
                        "device_laplacian_build_ms": dev.build_ms,
                        "estimate_lmax_bounds": rw.t_lmax_bounds, "estimate_lmax_lanczos_device": t_lanczos,
                        "lanczos_ritz_over_bound": lanczos_ratio, "placement_tuning": rw.tuning},
        }

    # ---- end to end through the mirrored API: numpy in -> Filter.filter -> numpy out (PCIe both ways,
    # coefficient quadrature, shape handling); reported beside the device-resident rate, never as `value`
    if rank == 0 and world == 1 and not a.no_e2e:
        # the default estimate of Graph.lmax (graph.py:858-931): Lanczos on the device, beside the bound (here, not
        # before the timed region: the profiling children of this script - which pass --no-e2e - must see the
        # recurrence's launches only)
        t0 = time.perf_counter()
        ritz, _ = dev.lanczos_lmax(max_iter=80, tol=5e-4)
        out["setup_s"]["estimate_lmax_lanczos_device"] = time.perf_counter() - t0
        out["setup_s"]["lanczos_ritz_over_bound"] = ritz * 1.01 / lmax
        # the generator class itself (nngraphs/sensor.py: points -> k-NN -> weights -> Graph.__init__): the builder's W
        # stays on the device and goes straight into the graph set-up; the host copy behind G.W is made on first access
        best_gen = None
        for _ in range(2):
            t0 = time.perf_counter()
            Gs = graphs.Sensor(N, k=a.knn, seed=42, compute_dtype=dtype, ctx=ctx)
            dt_gen = time.perf_counter() - t0
            best_gen = dt_gen if best_gen is None else min(best_gen, dt_gen)
            t0 = time.perf_counter()
            nnz_w = Gs.W.nnz
            t_w = time.perf_counter() - t0
            del Gs
        out["setup_s"]["sensor_generator_device_chain"] = best_gen
        out["setup_s"]["sensor_generator_first_access_of_W"] = t_w
        out["setup_s"]["sensor_generator_nnz_W"] = int(nnz_w)
        flt = filters.Heat(G, a.scale)
        flt.filter(x[:, :4], method="chebyshev", order=K)  # warm-up (allocations)
        flt.filter(x, method="chebyshev", order=K)         # ... and the pinned staging buffers of the pipeline
        best, stages = None, None
        for _ in range(3):
            te = time.perf_counter()
            y_host = flt.filter(x, method="chebyshev", order=K)
            dt_ = time.perf_counter() - te
            if best is None or dt_ < best:
                best, stages = dt_, ctx.last_host_timing()
        t_e2e = best
        ctx.set_option("host_pipeline", 0)  # the round-2 form beside it: one copy in, the kernels, one copy out
        flt.filter(x, method="chebyshev", order=K)
        te = time.perf_counter()
        flt.filter(x, method="chebyshev", order=K)
        t_one_shot = time.perf_counter() - te
        ctx.set_option("host_pipeline", 1)
        out["end_to_end_host_arrays"] = {
            "ms": t_e2e * 1e3, "value": N * nsig * K / t_e2e, "stages": stages,
            "one_shot_ms": t_one_shot * 1e3,
            "note": "pygsp_amd.filters.Heat(G, scale).filter(x_host, method='chebyshev', order=K), pageable numpy "
                    "arrays in and out: coefficient quadrature, shape handling, and gspx_cheby_filter - signal-column "
                    "batches pipelined over pinned staging (host threads pack | H2D | kernels | D2H | unpack, three "
                    "streams); stages = gspx_last_host_timing of the best of 3 calls; one_shot_ms = the same call with "
                    "option host_pipeline=0 (one pageable copy in, the kernels, one copy out)"}
        assert y_host.shape == (N, nsig)
        del y_host
        step()  # leave the timed path's result in the output buffer for the parity check below
        fence()

    # ---- the same headline workload in float32 (north_star's 1e-3 mode), same graph object --
    if rank == 0 and world == 1 and a.dtype == "f64" and not a.no_f32:
        out["headline_f32"] = headline_other_dtype(a, ctx, G, c, x, lmax, np.float32, oracle=not a.no_cpu)
    # ---- three filters chained through the drop-in API, device-resident against numpy in / numpy out ----------
    if rank == 0 and world == 1 and not a.no_e2e and not a.no_chain:
        try:
            out["chain3"] = chain3(a, G, x, K, oracle=not a.no_cpu)
        except Exception as e:  # an extra: never a reason to lose the measurement
            out["chain3"] = {"error": repr(e)}
        step()
        fence()

    # ---- CPU baseline beside it (rank 0, N=1 only): the oracle on a bounded column sample --------
    if rank == 0 and not a.no_cpu:
        from oracle import cheby_oracle as orc
        cols = min(a.cpu_cols, nsig)
        L = G.L.astype(np.float64)
        xs = x[:, :cols].astype(np.float64)
        orc.cheby_op(L, lmax, c[0], xs[:, :1])  # warm-up
        tc = time.perf_counter()
        ref = orc.cheby_op(L, lmax, c[0], xs)
        t_cpu = time.perf_counter() - tc
        y = download_y()
        err = float(np.max(np.abs(y[:, :cols] - ref)) / np.max(np.abs(ref)))
        out["cpu_baseline"] = {
            "value": N * cols * K / t_cpu, "unit": "vertex*signal*order/s", "cores": 1,
            "kind": "port",
            "sample": "oracle port of the reference (scipy csr_matvecs, single-threaded like the reference): same "
                      "graph/coefficients, first {} of {} signal columns, order {}, float64 ({} host cores "
                      "present), {:.1f} s".format(cols, nsig, K, os.cpu_count(), t_cpu),
            "note": CPU_BASELINE_NOTE,
            "multi_core": cpu_all,
        }
        # the reference ITSELF beside it whenever a real pygsp can be imported on this box (installed,
        # or $PYGSP_PATH - /root/reference does not travel to the driver's GPU box, so the driver's line says "port")
        ref_run = reference_cpu_baseline(W, lmax, a.scale, K, xs)
        if ref_run is not None:
            t_ref, y_ref, where = ref_run
            out["cpu_baseline"].update(
                kind="reference", value=N * cols * K / t_ref, port_value=N * cols * K / t_cpu,
                sample="pygsp {} itself ({}): filters.Heat(G, {:g}).filter(x[:, :{}], method='chebyshev', order={}) "
                       "on the same W and lmax, float64, one core, {:.1f} s; port_value = the oracle port on the same "
                       "sample ({:.1f} s)".format(where[0], where[1], a.scale, cols, K, t_ref, t_cpu),
                reference_vs_port_max_rel_diff=float(np.max(np.abs(y_ref - ref)) / np.max(np.abs(ref))))
            out["parity_vs_reference"] = {"max_rel_err": float(np.max(np.abs(y[:, :cols] - y_ref)) / np.max(np.abs(y_ref))),
                                          "columns": cols, "tolerance": 1e-5 if a.dtype == "f64" else 1e-3}
        out["parity_vs_oracle"] = {"max_rel_err": err, "columns": cols,
                                   "tolerance": 1e-5 if a.dtype == "f64" else 1e-3}
        for key_, y_form in (("newton_form", y_newton), ("product_form", y_product)):  # the other evaluations of the
            if y_form is not None and out.get(key_):                                  # same call, same oracle columns
                out[key_]["parity_vs_oracle"] = {
                    "max_rel_err": float(np.max(np.abs(y_form[:, :cols] - ref)) / np.max(np.abs(ref))), "columns": cols,
                    "tolerance": 1e-5 if a.dtype == "f64" else 1e-3}
    # ---- the other BASELINE configs (N=1 only), appended after the headline keys ---------------------
    if rank == 0 and world == 1 and not a.no_configs:
        bx.free()
        if by is not None:
            by.free()
        for g_ in list(G._dev.values()):
            g_.destroy()
        G._dev = {}
        try:
            out["configs"] = run_configs(ctx, a.only_config, a.config_reps, a.config_oracle_cols)
        except Exception as e:
            out["configs"] = {"error": repr(e)}
    if rank == 0 and parity_multi is not None:
        out["parity_vs_oracle"] = parity_multi
    if rank == 0:
        # the numbers a reader of the driver's record needs sit INSIDE `roofline` (the driver keeps that dict whole and
        # only the names of the other extra keys): the same bytes over the whole call, the bytes measured over the
        # bytes counted, the fp32 headline, the Newton form, the parity, the other configs' fractions
        rf = out["roofline"]
        rf["frac_whole_call"] = b_alg_call / (out["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        rf["traffic_over_algorithmic"] = (traffic / b_alg_launch) if traffic else None
        par = out.get("parity_vs_oracle") or {}
        rf["parity_max_rel_err"], rf["parity_tolerance"] = par.get("max_rel_err"), par.get("tolerance")
        f32 = out.get("headline_f32") or {}
        rf["f32_frac"] = (f32.get("roofline") or {}).get("frac")
        rf["f32_frac_whole_call"] = f32.get("frac_whole_call")
        rf["f32_parity_max_rel_err"] = (f32.get("parity_vs_oracle") or {}).get("max_rel_err")
        rf["f32_frac_of_mix_ceiling"] = (f32.get("roofline") or {}).get("frac_of_mix_ceiling")
        rf["f32_mix_ceiling_frac"] = (f32.get("roofline") or {}).get("mix_ceiling_frac")
        rf["f32_auto_evaluation"] = f32.get("auto_evaluation")
        rf["f32_auto_frac"] = ((f32.get("newton_form") or {}).get("frac_of_8TBps")
                               if f32.get("auto_evaluation") in ("newton", "product") else rf["f32_frac"])
        rf["f32_newton_parity_max_rel_err"] = ((f32.get("newton_form") or {}).get("parity_vs_oracle") or {}).get("max_rel_err")
        nf_ = out.get("newton_form") or {}
        rf["newton_frac"] = nf_.get("frac_of_8TBps")
        rf["newton_parity_max_rel_err"] = (nf_.get("parity_vs_oracle") or {}).get("max_rel_err")
        # what evaluation='auto' (plugin.install(evaluation='auto') / Filter.filter(..., evaluation='auto')) runs for this
        # very call, and the fraction it reaches: the Newton form when the guard clears it, else the recurrence
        pf_ = out.get("product_form") or {}
        rf["product_frac"] = pf_.get("frac_of_8TBps")
        rf["product_parity_max_rel_err"] = (pf_.get("parity_vs_oracle") or {}).get("max_rel_err")
        rf["product_panel_passes_per_order"] = (pf_.get("guard") or {}).get("panel_passes_per_order")
        rf["auto_evaluation"] = nf_.get("auto_picks")
        rf["auto_frac"] = (rf["product_frac"] if nf_.get("auto_picks") == "product" else
                           rf["newton_frac"] if nf_.get("auto_picks") == "newton" else (rf["frac"] if nf_ else None))
        rf["auto_parity_max_rel_err"] = (rf["product_parity_max_rel_err"] if nf_.get("auto_picks") == "product" else
                                         rf["newton_parity_max_rel_err"] if nf_.get("auto_picks") == "newton" else
                                         rf["parity_max_rel_err"])
        # the mix ceiling: what THIS box's memory system delivers to the step's own access mix with the arithmetic
        # removed (same launches, same bytes).  frac_of_mix_ceiling = mix time / step time of calls alternating in the
        # same minute (1.0: the step is bound by the memory system serving this mix, not by its row products);
        # mix_ceiling_GBps = the step's measured HBM bytes per launch over the mix time; mix_ceiling_frac = `frac` if
        # the step ran at the mix kernel's speed; the *_nobarrier figures drop the two workgroup barriers of a pass too
        if mix is not None and "error" not in mix:
            rf["read_GBps_this_run"] = mix["read_GBps"]
            rf["mix_launch_ms"], rf["mix_nobarrier_launch_ms"] = mix[1], mix[2]
            rf["step_launch_ms_beside_mix"] = mix["real"]
            rf["frac_of_mix_ceiling"] = mix[1] / mix["real"]
            rf["frac_of_mix_ceiling_nobarrier"] = mix[2] / mix["real"]
            rf["mix_ceiling_GBps"] = (traffic / (mix[1] * 1e-3) / 1e9) if traffic else None
            rf["step_traffic_GBps"] = (traffic / (mix["real"] * 1e-3) / 1e9) if traffic else None
            rf["mix_ceiling_frac"] = b_alg_launch / (mix[1] * 1e-3) / 1e9 / HBM_PEAK_GBS
            rf["mix_nobarrier_ceiling_frac"] = b_alg_launch / (mix[2] * 1e-3) / 1e9 / HBM_PEAK_GBS
            rf["smi_under_load"] = mix.get("smi") or None
            rf["stream_mix_GBps"] = mix.get("streams")
            tun = (out.get("setup_s") or {}).get("placement_tuning") or {}
            rf["placement_candidates_frac"] = tun.get("candidates_frac")
            rf["frac_untuned_first_draw"] = (tun.get("candidates_frac") or [None])[0]
        elif mix is not None:
            rf["mix_error"] = mix["error"]
        if spread is not None:
            rf.update(spread)
        cfgs = out.get("configs")
        if isinstance(cfgs, list):
            rf["configs_frac"] = {"{}_{}".format(c_.get("key", i), c_.get("dtype", "")): (c_.get("roofline") or {}).get("frac")
                                  for i, c_ in enumerate(cfgs) if isinstance(c_, dict)}
    if rank == 0 and batch5 is not None:
        out["batch_config4"] = batch5
    # C-level stdio first (RCCL prints a version banner through printf when a communicator is created; on a
    # pipe it would otherwise come out at process exit, after the result): the JSON line is the last line
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(out), flush=True)
    if torch is not None:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
