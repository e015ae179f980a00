#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: leapfrog-steps/s (HMC) + chain-steps/s (MCMC) at 65 536
chains per GPU, with SMC and the coupled-model configurations beside them.

Headline workload (BASELINE.json configs[1] + north_star): `hmc_chain` on the 32-site conjugate Normal model
(x#i ~ N(0,1); y#i ~ N(x#i, 0.5) observed at 0.2 i - 1), 65 536 chains per GPU, HMCConfig::default() (L = 16, h = 1e-5,
target accept 0.8), gradient mode FG_GRAD_FD_SPARSE -- the engine's default: the reference's central difference
(hmc.rs:304-329) over the statements that read the perturbed coordinate.  The reference-verbatim dense mode (2 d whole-model
runs per gradient) is a leg of its own (`hmc_fd_dense`, with its roofline) and is what the CPU baseline is compared with.

A bench "step" = ONE HMC transition (16 leapfrog steps, 17 gradients, endpoint score, accept/reject, adaptation) of EVERY
chain.  `--warmup W` untimed transitions are the adaptive warmup; then the K-transition timed region (`--steps K`, each
bracketed by barrier + synchronize, MAX over ranks) runs `--repeats R` times back to back on the same engine into
consecutive draw slabs [K][d][C] in HBM: `value` = world x chains x K x L / MEDIAN region time, `ms_per_step` = that
median / K, the R values beside them (`timed_regions`).

`--scaling weak` (default): every rank runs `--chains` chains (65 536).  `--scaling strong`: the job is `--chains` chains in
total (BASELINE's C3 / C5 are fixed-size jobs sharded 8 x), split evenly over the ranks.  Either way chains are keyed by their
global id (Philox), there is no data-path collective, and the only exchange is the R-hat / ESS all-reduce after the timed
region, inside the library (fg_diag_rhat_ess: O(d) doubles per rank).

`--gpus N` without a torchrun environment starts the N ranks itself (a `python -m torch.distributed.run` child, before
anything here touches HIP) and relays rank 0's line.

One JSON line on stdout (rank 0), at most ~5 KB -- the driver's record keeps the contract keys, `roofline`, `cpu_baseline` and a tail:
  roofline      the dominant kernel (as reported by the engine: fg_hmc_last_kernel): algorithmic log-pdf evaluations x 8 flops /
                HIP-event time against the f64 vector peak; executed flops / VALU issue / `traffic` from the committed rocprofv3 --pmc
                passes of the same configuration (profiles/roundN_pmc.json, newest first; `executed_source` says which entry).
  cpu_baseline  the CPU oracle (C restatement of the reference algorithm, dense FD) on this box's host cores, bounded sample.
  check, validity   statistics of the timed draws; a fixed 200 + 200 run of the headline model at 65 536 chains:
                |pooled mean - closed form| <= 1e-3, split R-hat.
  legs          LAST in the line: value / roofline.frac / kernel / cpu baseline of every other half of BASELINE's metric --
                mh (+ mh_8192), c5 (+ c5_32768), smc, c3_65536, c3_8192, hmc_fd_dense (the reference's arithmetic verbatim; the only
                GPU figure the CPU baseline may be divided into), hmc_8192.
The verbose document (every leg's notes, spreads, sub-objects) goes to --full-out (default gpurun_out/bench_full.json) and, with
--full, to stdout instead of the compact line; the committed copy is profiles/roundN_bench_full.json.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

N_SITES = 32
CHAINS_PER_GPU = 65536
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)
F64_VALU_PEAK_TFLOPS = 78.6    # 256 CU x 4 SIMD x 16 lanes x 2 flop (FMA) x 2.4 GHz
FLOPS_PER_NORMAL_LOGPDF = 8.0  # a Normal log-pdf with ln(sigma) hoisted ~ 8 flops
SMC_PARTICLES = 1 << 20
C3_N, C3_P = 1024, 32
EXIT_DIAGNOSTICS_FAILED = 3


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--repeats", type=int, default=5, help="how many times the K-transition timed region runs (median = value)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: --chains per rank; strong: --chains in total, sharded over the ranks")
    ap.add_argument("--chains", type=int, default=CHAINS_PER_GPU, help="chains per GPU (weak) / in total (strong)")
    ap.add_argument("--grad", choices=["fd_sparse", "fd_dense", "analytic"], default="fd_sparse",
                    help="fd_sparse (engine default) / fd_dense (reference verbatim): the reference's central difference; "
                         "analytic: closed-form derivative (not the reference's arithmetic)")
    ap.add_argument("--launch", type=int, default=25, help="transitions fused per kernel launch")
    ap.add_argument("--leapfrog", type=int, default=16, help="L (HMCConfig::default is 16; other values are for experiments only)")
    ap.add_argument("--spinup", type=float, default=0.4, help="seconds of untimed throw-away transitions on a scratch engine before the measured "
                    "engine starts (the GPU's clocks settle over the first tens of ms of f64 load); 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline leg only (no MH / SMC / C3 / C5 / dense / validity legs)")
    ap.add_argument("--full", action="store_true", help="print the verbose document (every leg's notes, spreads and sub-objects) instead of the compact line")
    ap.add_argument("--full-out", default=os.environ.get("FG_BENCH_FULL_OUT", os.path.join("gpurun_out", "bench_full.json")),
                    help="where the verbose document is written beside the compact line ('' = nowhere)")
    ap.add_argument("--cpu-chains", type=int, default=4096)
    ap.add_argument("--cpu-transitions", type=int, default=64)
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------ rank launcher
def launch_ranks(args) -> int:
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a child process group and relay rank 0's JSON
    line.  Nothing in THIS process has imported torch or touched HIP (a process that has initialised the GPU must not
    exec or fork GPU children on this pool)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
    line = None
    for ln in p.stdout:
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if line is not None:
        print(line, flush=True)
    if rc != 0:
        sys.stderr.write(f"bench.py: the {args.gpus}-rank child exited with code {rc}\n")
        return rc
    if line is None:
        sys.stderr.write("bench.py: the ranks produced no result line\n")
        return 1
    return 0


# ------------------------------------------------------------------------------------------ CPU baselines (rank 0, N = 1)
def host_cores() -> int:
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:                                                   # a container's CPU quota, when it is tighter than the affinity mask
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return cores


def cpu_baseline_hmc(args, gpu_dense):
    """The CPU oracle (oracle/: per-chain sequential, interpretive, dense central FD exactly as hmc.rs:304-329) on a
    bounded sample of the same workload, all host cores."""
    from fugue_amd import workloads as W
    from oracle import oracle as orc
    om = orc.OracleModel(W.normal_sites(N_SITES))
    cores = host_cores()
    nw = args.cpu_transitions // 2
    ns = args.cpu_transitions - nw
    t0 = time.perf_counter()
    om.hmc_run(1, args.cpu_chains, nw, ns, orc.HmcConfig.default(), n_threads=cores, want_draws=False)
    dt = time.perf_counter() - t0
    lf = args.cpu_chains * args.cpu_transitions * 16
    out = {"value": lf / dt, "unit": "leapfrog-steps/s", "cores": cores, "kind": "port",
           "sample": f"{args.cpu_chains} chains x {args.cpu_transitions} transitions (L=16, dense FD = hmc.rs:304-329 verbatim) of the same "
                     f"model, {dt:.1f} s wall; C restatement, not the Rust binary",
           "published_reference": "none for HMC (BASELINE.md: derived ~1.6e3 leapfrog-steps/s/thread at d=20)"}
    if gpu_dense:
        out["like_for_like"] = {"gpu_fd_dense_leapfrog_steps_per_sec": gpu_dense, "ratio": gpu_dense / out["value"],
                                "note": "the oracle has only the reference's dense FD; compare it with the GPU's dense mode (hmc_fd_dense), not with the sparse headline"}
    return out


def cpu_baseline_c3(cores):
    """The oracle on C3 (dense FD: 64 whole-model runs of 1 056 statements per gradient), bounded sample."""
    from fugue_amd import workloads as W
    from oracle import oracle as orc
    X, y, _ = W.ridge_data(C3_N, C3_P)
    om = orc.OracleModel(W.ridge_regression(X, y))
    chains, nt = 8 * cores, 2
    t0 = time.perf_counter()
    om.hmc_run(1, chains, 0, nt, orc.HmcConfig.default(init_step_size=0.004), n_threads=cores, want_draws=False)
    dt = time.perf_counter() - t0
    return {"value": chains * nt * 16 / dt, "unit": "leapfrog-steps/s", "cores": cores, "kind": "port",
            "sample": f"{chains} chains x {nt} transitions (L=16, fixed step 0.004, dense FD = hmc.rs:304-329 verbatim) of the same model, {dt:.1f} s wall; "
                      "C restatement, not the Rust binary"}


def cpu_baseline_mh():
    """Single-thread calibration of the restatement against the reference's published criterion numbers
    (benches/f_perf.rs:24-28: 50 + 50 transitions, reference_model(20) 15.3 us and reference_model(50) 73.1 us per
    transition on Apple Silicon), then all cores on the bench model."""
    from fugue_amd import workloads as W
    from oracle import oracle as orc
    cal = {}
    for n_sites, published_us in ((20, 15.3), (50, 73.1)):
        om = orc.OracleModel(W.reference_model(n_sites))
        chains = 20000 if n_sites == 20 else 8000
        t0 = time.perf_counter()
        om.mh_run(1, chains, 50, 50, None, [0], n_threads=1, want_draws=False)
        dt = time.perf_counter() - t0
        cal[f"reference_model({n_sites})"] = {"us_per_transition_1_thread": dt / (chains * 100) * 1e6, "published_rust_us_per_transition": published_us,
                                              "sample": f"{chains} chains x (50 + 50) transitions, 1 thread"}
    cores = host_cores()
    om = orc.OracleModel(W.reference_model(20))
    chains = 8192 * max(1, cores // 4)
    t0 = time.perf_counter()
    om.mh_run(1, chains, 1000, 1000, None, [0], n_threads=cores, want_draws=False)
    dt = time.perf_counter() - t0
    return {"value": chains * 2000 / dt, "unit": "chain-steps/s", "cores": cores, "kind": "port",
            "sample": f"{chains} chains x (1000 + 1000) steps of reference_model(20), {dt:.1f} s wall; C restatement, not the Rust binary",
            "calibration": cal}


def cpu_baseline_smc(n=1 << 20):
    from fugue_amd import workloads as W
    from oracle import oracle as orc
    om = orc.OracleModel(W.smc_normal())
    t0 = time.perf_counter()
    r = om.smc_run(n, 42, method=1, ess_threshold=0.5, rejuvenation_steps=3, batched=0)
    dt = time.perf_counter() - t0
    moves = (r["n_model_evals"] - n) / 2
    return {"value": moves / dt, "unit": "particle-moves/s", "cores": 1, "kind": "port",
            "adaptation": "sequential: one shared DiminishingAdaptation updated after every particle (smc.rs:482,544-553) -- the reference's form",
            "sample": f"adaptive_smc, {n} particles, Systematic / 0.5 / 3 rejuvenation moves, {len(r['betas'])} tempering steps, {dt:.2f} s wall; "
                      "sequential by construction; C restatement, not the Rust binary",
            "seconds_scaled_to_1048576_particles": dt * (SMC_PARTICLES / n)}


# ------------------------------------------------------------------------------------------ profile lookups
PMC_FILES = ("round4_pmc.json", "round3_pmc.json")       # the newest committed set that holds the entry is used


def pmc_entry(key):
    """The committed rocprofv3 --pmc entry of one configuration (profiles/round3_pmc.json, written by tools/prof_round3_collect.py):
    per-transition (or per-step / per-run) counter values of the dominant kernel and FETCH_SIZE / WRITE_SIZE bytes.  Returns
    (entry or None, source string)."""
    why = []
    for name in PMC_FILES:
        try:
            doc = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception as ex:                                                # noqa: BLE001
            why.append(f"no profiles/{name} ({type(ex).__name__})")
            continue
        ent = doc.get("entries", {}).get(key)
        if ent is not None:
            return ent, f"profiles/{name}[{key}]"
        why.append(f"profiles/{name} has no entry `{key}`")
    return None, "; ".join(why) + " (this run is not a profiled configuration)"


def executed_from_pmc(ent, units, seconds):
    """What the kernel actually issues, from the committed PMC counts (per unit) over this run's time: executed f64 flops
    (add / mul = 1, fma = 2 per lane) and the share of SIMD cycles that issue a VALU instruction (a wave64 VALU instruction
    holds its SIMD for 4 cycles)."""
    m = ent.get("counters_per_unit") if ent else None
    if not m or "SQ_INSTS_VALU" not in m:
        return None
    add, mul, fma = m.get("SQ_INSTS_VALU_ADD_F64", 0.0), m.get("SQ_INSTS_VALU_MUL_F64", 0.0), m.get("SQ_INSTS_VALU_FMA_F64", 0.0)
    flops = 64.0 * (add + mul + 2.0 * fma) * units
    simd_cycles = seconds * 2.1e9 * 1024                                       # 256 CUs x 4 SIMDs at the ~2.1 GHz the clock holds under f64 load
    return {"f64_tflops": flops / seconds / 1e12, "f64_share_of_valu": (add + mul + fma) / m["SQ_INSTS_VALU"],
            "valu_issue_share_of_simd_cycles_at_2.1GHz": 4.0 * m["SQ_INSTS_VALU"] * units / simd_cycles,
            "wave_instructions_per_unit": m["SQ_INSTS_VALU"], "unit": ent.get("unit", "transition")}


def traffic_from_pmc(ent, units):
    if not ent or "fetch_bytes_per_unit" not in ent:
        return None
    return (ent["fetch_bytes_per_unit"] + ent["write_bytes_per_unit"]) * units


NATIVE_RCCL_TIMEOUT_S = float(os.environ.get("FG_BENCH_RCCL_TIMEOUT", "120"))
T_START = time.perf_counter()


def progress(rank, msg):
    """One stderr line per finished leg (stdout carries the single JSON line)."""
    sys.stderr.write(f"[bench rank {rank} +{time.perf_counter() - T_START:.1f}s] {msg}\n")
    sys.stderr.flush()


def call_with_timeout(fn, seconds):
    """fn() on a daemon thread (ctypes calls release the GIL); raises TimeoutError when it has not returned in time -- the
    thread is then abandoned: the caller finishes its line from rank-local data and the process ends with os._exit
    (non-zero), WITHOUT entering another collective on this device."""
    import threading
    box = {}

    def run():
        try:
            box["r"] = fn()
        except BaseException as ex:                              # noqa: BLE001
            box["e"] = ex

    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        ABANDONED.append(th)
        raise TimeoutError(f"no return within {seconds:.0f} s")
    if "e" in box:
        raise box["e"]
    return box.get("r")


ABANDONED = []


class Clock:
    """Barrier + synchronize on both sides of a region; MAX over ranks."""

    def __init__(self, torch, dist, world, coll_dev):
        self.torch, self.dist, self.world, self.coll_dev = torch, dist, world, coll_dev

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def region(self, fn):
        self.barrier()
        t0 = time.perf_counter()
        fn()
        self.barrier()
        dt = time.perf_counter() - t0
        if self.world > 1:
            t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.coll_dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt


def stepped(torch, stream, step_fn, total, per_launch):
    """Runs `total` steps in launches of `per_launch`, each bracketed by HIP events on the engine's stream."""
    events, done = [], 0
    while done < total:
        n = min(per_launch, total - done)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        step_fn(n, done)
        e1.record(stream)
        events.append((e0, e1, n))
        done += n
    return events


def full_launch_ms(events):
    ms = [(e0.elapsed_time(e1), n) for e0, e1, n in events]
    n0 = events[0][2]
    return float(np.mean([t for t, n in ms if n == n0])), n0


def spread(vals):
    v = sorted(vals)
    return {"median": float(np.median(v)), "min": v[0], "max": v[-1], "all": list(vals)}


def shard(total_or_per_rank, world, scaling):
    if scaling == "weak":
        return total_or_per_rank
    if total_or_per_rank % (64 * world):
        raise SystemExit(f"--scaling strong: {total_or_per_rank} chains do not split into whole 64-chain tiles over {world} ranks")
    return total_or_per_rank // world


class Ctx:
    pass


# ------------------------------------------------------------------------------------------ one rank
def run_rank(args):
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or run `python bench.py --gpus N` outside torchrun and let it start the ranks)")
    # FG_BENCH_ONE_DEVICE=1 rehearses the N > 1 code path on a one-GPU box: every rank drives cuda:0 and the
    # collectives run over gloo on host tensors (RCCL refuses two ranks on one device).  Not a measurement mode.
    one_device = os.environ.get("FG_BENCH_ONE_DEVICE", "0") == "1"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl" if (torch.cuda.is_available() and not one_device) else "gloo")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = "cpu" if one_device else f"cuda:{local_rank}"
    clock = Clock(torch, dist, world, coll_dev)

    from fugue_amd import engine as E, workloads as W
    from fugue_amd import diagnostics as D
    X = Ctx()
    X.args, X.E, X.W, X.D, X.torch, X.dist, X.clock, X.world, X.rank, X.dev = args, E, W, D, torch, dist, clock, world, rank, local_rank
    X.stream = torch.cuda.current_stream()
    X.coll_dev, X.one_device = coll_dev, one_device
    C, K, Wn, L, R = shard(args.chains, world, args.scaling), args.steps, args.warmup, args.leapfrog, max(1, args.repeats)
    cp = E.compile_model(W.normal_sites(N_SITES))
    d = cp.d
    mode = {"fd_sparse": E.GRAD_FD_SPARSE, "fd_dense": E.GRAD_FD_DENSE, "analytic": E.GRAD_ANALYTIC}[args.grad]
    cfg = E.hmc_config(grad_mode=mode, n_leapfrog=L)
    eng = E.Engine(cp, C, seed=1, chain_offset=rank * C, device=local_rank)
    stream = X.stream
    eng.set_stream(stream.cuda_stream)                    # kernels + torch events share one stream
    slab = K * d * C * 8
    R = max(1, min(R, int(100e9 // max(1, slab))))        # consecutive draw slabs of the R timed regions (<= 100 GB of the 288 GB)
    draws = torch.empty((R, K, d, C), dtype=torch.float64, device=f"cuda:{local_rank}")

    eng.hmc_init(cfg, Wn)
    draws.zero_()                                         # the output buffer's pages are touched before anything is timed
    # ---- clock spin-up first: throw-away transitions of the same kernel on a scratch engine (own state, own seed; the measured
    # engine is not touched), so that the W warmup and the timed transitions run at the clocks the GPU settles to under this
    # load rather than on its way there.  The measured engine's W warmup transitions then run directly before the timed regions.
    scratch = None
    if args.spinup > 0:
        scratch = E.Engine(cp, C, seed=987654321, chain_offset=rank * C, device=local_rank)
        scratch.set_stream(stream.cuda_stream)
        scratch.hmc_init(cfg, 0)
        t_sp = time.perf_counter()
        while time.perf_counter() - t_sp < args.spinup:
            scratch.hmc_step(4 * args.launch)
            torch.cuda.synchronize()
    # ---- untimed by the contract (reported separately): W adaptive warmup transitions
    t_warm = clock.region(lambda: stepped(torch, stream, lambda n, done: eng.hmc_step(n), Wn, args.launch) if Wn > 0 else None)
    # ---- timed: R regions of exactly K sampling transitions each, into consecutive draw slabs
    events, dts = [], []
    for r in range(R):
        ev = []
        dts.append(clock.region(lambda: ev.extend(stepped(torch, stream, lambda n, done: eng.hmc_step(n, draws[r, done].data_ptr()), K, args.launch))))
        events.extend(ev)
    kernel = eng.hmc_last_kernel()
    launch_ms, n_launch = full_launch_ms(events)
    if scratch is not None:
        scratch.close()
    dt = float(np.median(dts))
    last = draws[R - 1]

    # ---- after the timed regions: the ONLY cross-chain step -- split R-hat / multichain ESS of the last slab.  Each rank reduces
    # its own draws to chain sums on its GPU; the library all-reduces 6 d + 2 d doubles (+ 32 d per chunk of lags) over RCCL / xGMI
    # (fg_diag_rhat_ess; communicator created from an id that rank 0 obtains and torch.distributed's store hands out).
    t_diag = time.perf_counter()
    diag_path, comm, failed, exch, fallback_note = "library (single GPU)", None, False, 0, ""
    r_ = None
    if world == 1:
        r_ = eng.diag_rhat_ess(last.data_ptr(), K, d, None)
    elif not one_device or os.environ.get("FG_BENCH_FORCE_NATIVE_RCCL") == "1":    # (forced in the rehearsal mode: exercises the error path)
        native_err = None
        try:
            ids = [E.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            if os.environ.get("FG_BENCH_FAKE_RCCL_HANG") == "1":     # rehearsal of the watchdog: a communicator that never forms
                call_with_timeout(lambda: time.sleep(3600), 2.0)
            # under a watchdog: a communicator that cannot form must cost a bounded wait, not the run
            comm = call_with_timeout(lambda: eng.comm_init(world, rank, ids[0]), NATIVE_RCCL_TIMEOUT_S)
            r_ = call_with_timeout(lambda: eng.diag_rhat_ess(last.data_ptr(), K, d, comm), NATIVE_RCCL_TIMEOUT_S)
            call_with_timeout(lambda: E.comm_destroy(comm), NATIVE_RCCL_TIMEOUT_S)
            diag_path = "library: ncclAllReduce of chain sums (6 d + 2 d + 32 d per lag chunk doubles)"
            exch = r_["exchange_bytes"]
        except BaseException as ex:                              # noqa: BLE001
            native_err, r_ = ex, None
        # Two kinds of failure.  A call that did NOT RETURN (watchdog): a thread of this process still sits inside RCCL on this
        # device, so no further collective of any kind -- the line is finished from this rank's own chains and the process exits
        # non-zero (main).  A call that RETURNED an error (the library could not bind / initialise RCCL): nothing is stuck, the
        # ranks agree on it through torch.distributed (itself under the watchdog: a peer that hung never joins) and take the
        # same O(d) exchange through torch.distributed's all-reduce instead; the line says so and the exit code stays 0.
        hung = bool(ABANDONED)
        if not hung:
            try:
                ok = torch.tensor([0 if native_err is not None else 1], dtype=torch.int32, device="cpu" if one_device else coll_dev)
                call_with_timeout(lambda: (dist.all_reduce(ok, op=dist.ReduceOp.MIN), ok.cpu()), NATIVE_RCCL_TIMEOUT_S)
                if int(ok.cpu()[0]) == 0:
                    r_ = None
                    if native_err is None:
                        native_err = RuntimeError("another rank's library exchange returned an error")
            except BaseException as ex:                          # noqa: BLE001
                hung, native_err = True, native_err or ex
        if hung:
            sys.stderr.write(f"rank {rank}: the library's RCCL exchange did not return ({native_err!r}); finishing from rank-local data, exit code {EXIT_DIAGNOSTICS_FAILED}\n")
            failed, r_ = True, None
            diag_path = f"failed ({type(native_err).__name__}: {native_err}); R-hat / ESS below are of THIS rank's chains only"
        elif native_err is not None:
            sys.stderr.write(f"rank {rank}: the library's RCCL exchange returned an error ({native_err!r}); taking the same exchange through torch.distributed\n")
            fallback_note = f" (the library's own RCCL exchange returned an error: {type(native_err).__name__}: {native_err})"
    if r_ is None and not failed and world > 1:                  # the same reduce exchange over torch.distributed (one-device rehearsal: gloo)
        prov = D.EngineMoments(eng, last.data_ptr(), K, d)
        cd = D.ChainDiagnostics(prov, device=None if one_device else coll_dev)
        r_ = dict(r_hat=cd.split_rhat(), ess=cd.ess(), chains=cd.m)
        exch = cd.exchange_bytes
        prov.close()
        diag_path = "torch.distributed all-reduce of chain sums + library combination" + (fallback_note or " (rehearsal mode)")
    if r_ is None:                                               # failure path: rank-local statistics, no collective
        r_ = eng.diag_rhat_ess(last.data_ptr(), K, d, None)
    rhat, ess, n_chains_diag = r_["r_hat"], (r_["ess"] if K >= 4 else np.full(d, float("nan"))), r_["chains"]
    t_diag = time.perf_counter() - t_diag

    st = eng.hmc_stats()
    m = last.mean(dim=(0, 2)).cpu().numpy()
    v = last.var(dim=(0, 2)).cpu().numpy()
    _, tm, tv = W.normal_sites_truth(N_SITES)
    mean_err, var_err = float(np.abs(m - tm).max()), float(np.abs(v - tv).max())
    eng.close()
    del draws, last

    value = world * C * K * L / dt
    # ---- roofline of the dominant kernel: algorithmic f64 work / HIP-event time
    n_stmt = 2 * N_SITES                                                      # S + O statements of the model
    evals_sparse = (2 * d * (L + 1)) * 2 + n_stmt                             # 2 signs x 2 dependent statements per coordinate per gradient + endpoint score
    evals_dense = (2 * d * (L + 1)) * n_stmt + n_stmt                         # 2 d (S + O) per gradient
    evals_dense_executed = (L + 1) * (n_stmt + 2 * 2 * d)                    # the dense kernel evaluates every statement once per gradient + the moved ones at +-h; the rest of the two scoring runs is additions
    evals = {E.GRAD_FD_SPARSE: evals_sparse, E.GRAD_FD_DENSE: evals_dense_executed, E.GRAD_ANALYTIC: d * (L + 1) * 2 + n_stmt}[mode]
    achieved_tflops = C * n_launch * evals * FLOPS_PER_NORMAL_LOGPDF / (launch_ms * 1e-3) / 1e12
    alg_bytes_per_launch = C * n_launch * (L * 32 * d + 8 * d + 16)           # 32 d B / leapfrog step (+ draw row, lj, eps)
    ent, src = pmc_entry(f"hmc|normal32|{C}|{args.grad}|L{L}")
    out = {
        "metric": "hmc_leapfrog_steps_per_sec", "value": value, "unit": "leapfrog-steps/s", "n_gpus": world,
        "steps": K, "warmup": Wn, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": args.scaling,
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "timed_regions": {"repeats": R, "steps_each": K, "seconds": spread(dts), "value": spread([world * C * K * L / t for t in dts]),
                          "note": "value / ms_per_step are those of the MEDIAN region; every region is K transitions between barrier + synchronize, MAX over ranks"},
        "config": {"workload": "C2-normal32: hmc_chain, 32-site conjugate Normal (x#i~N(0,1), y#i~N(x#i,0.5)=0.2i-1), "
                               f"{C} chains/GPU, L=16, HMCConfig::default", "chains_per_gpu": C, "chains_total": world * C, "n_sites": N_SITES,
                   "n_leapfrog": L, "grad": args.grad, "grad_note": "fd_sparse = the engine's default in the C ABI, Python and bench; the reference-verbatim dense "
                   "mode is the `hmc_fd_dense` leg", "transitions_per_launch": n_launch, "clock_spinup_seconds": args.spinup,
                   "sharding": (f"{args.scaling}: chains x{world}" if world > 1 else "single GPU")},
        "roofline": {"bound": "valu_f64", "achieved": achieved_tflops, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved_tflops / F64_VALU_PEAK_TFLOPS, "traffic": traffic_from_pmc(ent, n_launch), "traffic_source": src,
                     "kernel": kernel, "avg_launch_ms": launch_ms,
                     "logpdf_evals_per_transition": evals, "flops_per_logpdf": FLOPS_PER_NORMAL_LOGPDF,
                     "note": "achieved = log-pdf evaluations the launch performs x 8 flops / HIP-event time; peak = f64 vector FMA peak "
                             "(2 flops/instr at 2.4 GHz) -- the arithmetic is unfused add/mul (reference rounding, 1 flop/instr) and the clock "
                             "sits near 2.1 GHz under f64 load, so ~36 TFLOP/s is the ceiling of this instruction mix.  traffic = measured "
                             "FETCH_SIZE + WRITE_SIZE per launch (separate rocprofv3 --pmc passes)",
                     "executed": executed_from_pmc(ent, n_launch, launch_ms * 1e-3), "executed_source": src,
                     "dense_semantics": {"logpdf_evals_per_transition": evals_dense,
                                         "note": "SURVEY 8d's 2 d (S+O) log-pdfs per gradient are the two whole scoring runs of grad_log_joint; that arithmetic is the "
                                                 "`hmc_fd_dense` leg -- the sparse default never forms the terms that cancel in the reference's subtraction"},
                     "hbm_nominal": {"achieved": alg_bytes_per_launch / (launch_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "note": "SURVEY 8d algorithmic bytes (32 d B per leapfrog step, as if q and p round-tripped HBM) / time: NOT a "
                                             "claim -- q, p stay in registers for a whole trajectory and the kernel is not HBM bound"}},
        "incl_warmup": {"leapfrog_steps_per_sec": world * C * (K + Wn) * L / (dt + t_warm) if Wn > 0 else value,
                        "warmup_seconds": t_warm, "note": "SURVEY 8d counts leapfrog steps over warmup + sampling; `value` follows the bench contract (K timed "
                                                          "sampling transitions after W untimed warmup transitions)"},
        "check": {"posterior_mean_max_abs_err": mean_err, "posterior_var_max_abs_err": var_err,
                  "accept_rate": st.accept_rate, "mean_step_size": st.mean_step_size, "n_divergent": int(st.n_divergent),
                  "split_rhat_max": float(np.max(rhat)), "ess_min": float(np.min(ess)), "chains_in_rhat": int(n_chains_diag),
                  "diagnostics_seconds": t_diag, "diagnostics_path": diag_path, "diagnostics_exchange_bytes_per_rank": int(exch),
                  "note": "statistics of the K draws of the last timed region: with few steps / a short warmup they are NOT the 1e-3 evidence (a chain of "
                          "20 draws after 5 warmup transitions has not mixed) -- see `validity`"},
    }
    progress(rank, f"hmc leg done: {value:.4g} leapfrog-steps/s over {world} rank(s) [{kernel}]")
    if failed:
        # a thread of this process may still be inside RCCL: no further leg, no further collective
        out["check"]["note"] += "; the cross-rank exchange FAILED -- remaining legs skipped"
        if rank == 0:
            emit(args, out)
        return EXIT_DIAGNOSTICS_FAILED
    if not args.no_extras:
        out["mh"] = leg_mh(X)
        progress(rank, "mh leg done")
        out["smc"] = leg_smc(X)
        progress(rank, "smc leg done")
        out["c3"] = leg_c3(X)
        progress(rank, "c3 leg done")
        out["c5"] = leg_c5(X)
        progress(rank, "c5 leg done")
        if rank == 0:
            if args.scaling == "weak" and C != 8192:
                out["hmc_8192"] = leg_hmc_small(X, 8192)
                progress(rank, "hmc at 8192 chains done")
            out["hmc_fd_dense"] = leg_dense(X)
            progress(rank, "dense leg done")
            out["extras"] = extras(X)
            progress(rank, "extras done")
            out["validity"] = validity(E, W, D, local_rank)
            progress(rank, "validity leg done")
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            gpu_dense = out.get("hmc_fd_dense", {}).get("value")
            out["cpu_baseline"] = cpu_baseline_hmc(args, gpu_dense)
            if "hmc_fd_dense" in out:
                out["hmc_fd_dense"]["cpu_baseline"] = {k: out["cpu_baseline"][k] for k in ("value", "unit", "cores", "kind", "sample")}
                out["hmc_fd_dense"]["vs_cpu_baseline"] = gpu_dense / out["cpu_baseline"]["value"]
            if "mh" in out:
                out["mh"]["cpu_baseline"] = cpu_baseline_mh()
            if "smc" in out:
                out["smc"]["cpu_baseline"] = cpu_baseline_smc()
            if "c3" in out:
                out["c3"]["cpu_baseline"] = cpu_baseline_c3(host_cores())
        emit(args, out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


# ------------------------------------------------------------------------------------------ the line
def _sig(x, n=4):
    return float(f"{x:.{n}g}") if isinstance(x, float) and np.isfinite(x) else x


def _leg(d, extra=()):
    """One leg of the verbose document -> value, roofline fraction, kernel, CPU baseline."""
    if not d:
        return None
    rf, cb = d.get("roofline", {}), d.get("cpu_baseline") or {}
    o = {"value": _sig(d.get("value")), "unit": d.get("unit"), "frac": _sig(rf.get("frac"), 3), "bound": rf.get("bound"), "kernel": rf.get("kernel")}
    if "frac_kind" in rf:
        o["frac_kind"] = rf["frac_kind"].split(" -- ")[0][:60]
    if "reference_implied" in rf:
        o["reference_implied_frac"] = _sig(rf["reference_implied"]["frac"], 3)
    if rf.get("counter_traffic_frac") is not None:
        o["counter_traffic_frac"] = _sig(rf["counter_traffic_frac"], 3)
    ex = rf.get("executed") or {}
    if ex:
        o["executed_f64_tflops"] = _sig(ex.get("f64_tflops"), 3)
        o["valu_issue"] = _sig(ex.get("valu_issue_share_of_simd_cycles_at_2.1GHz"), 3)
        o["f64_share_of_valu"] = _sig(ex.get("f64_share_of_valu"), 3)
    if cb:
        o["cpu"] = {"value": _sig(cb.get("value")), "cores": cb.get("cores"), "kind": cb.get("kind")}
        if cb.get("value") and d.get("value"):
            o["vs_cpu"] = _sig(d["value"] / cb["value"], 3)
    for k in extra:
        if k in d:
            o[k] = _sig(d[k]) if isinstance(d[k], float) else d[k]
    return o


def compact(full):
    """The ONE line the driver records: the contract keys, `roofline` and `cpu_baseline` of the headline kernel, then -- LAST, so that a
    tail of the line still holds them -- `legs`: value / roofline fraction / kernel / CPU baseline of every other half of BASELINE's
    metric (mh, c5, smc, c3 at both chain counts, the reference-verbatim dense HMC).  Notes, spreads and sub-objects live in the verbose
    document (`--full`, written to --full-out; the committed copy is profiles/roundN_bench_full.json)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: full[k] for k in keep if k in full}
    cfg = full["config"]
    line["config"] = {"workload": cfg["workload"], "chains_per_gpu": cfg.get("chains_per_gpu"), "chains_total": cfg.get("chains_total"), "n_leapfrog": cfg.get("n_leapfrog"),
                      "grad": cfg["grad"] + (" (engine default: the reference's central difference over the statements that read the coordinate; "
                                             "reference-verbatim dense FD = legs.hmc_fd_dense)" if cfg["grad"] == "fd_sparse" else ""),
                      "transitions_per_launch": cfg.get("transitions_per_launch"), "sharding": cfg.get("sharding")}
    rf = full["roofline"]
    ex = rf.get("executed") or {}
    line["roofline"] = {"bound": rf["bound"], "achieved": _sig(rf["achieved"]), "peak": rf["peak"], "unit": rf["unit"], "frac": _sig(rf["frac"], 3),
                        "traffic": rf.get("traffic"), "kernel": rf["kernel"], "avg_launch_ms": _sig(rf["avg_launch_ms"]),
                        "logpdf_evals_per_transition": rf["logpdf_evals_per_transition"], "flops_per_logpdf": rf["flops_per_logpdf"],
                        "executed_f64_tflops": _sig(ex.get("f64_tflops"), 3), "valu_issue": _sig(ex.get("valu_issue_share_of_simd_cycles_at_2.1GHz"), 3),
                        "executed_source": rf.get("executed_source")}
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        line["cpu_baseline"] = {"value": _sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:120]}
        if "like_for_like" in cb:
            line["cpu_baseline"]["like_for_like_ratio_gpu_fd_dense"] = _sig(cb["like_for_like"]["ratio"], 3)
    line["timed_regions_value"] = [_sig(v) for v in full["timed_regions"]["value"]["all"]]
    ck = full["check"]
    line["check"] = {k: _sig(ck[k], 3) if isinstance(ck[k], float) else ck[k] for k in ("posterior_mean_max_abs_err", "accept_rate", "n_divergent", "split_rhat_max", "chains_in_rhat",
                                                                                       "diagnostics_path", "diagnostics_exchange_bytes_per_rank") if k in ck}
    line["check"]["diagnostics_path"] = str(line["check"].get("diagnostics_path", ""))[:200]
    line["check"] = dict({"of": f"the {full.get('steps')} timed draws behind {full.get('warmup')} warmup transitions from the prior draw (no convergence claim at this length: see validity)"}, **line["check"])
    if "validity" in full:
        v = full["validity"]
        line["validity"] = {"posterior_mean_max_abs_err": _sig(v["posterior_mean_max_abs_err"], 3), "target_1e-3": v["target_1e-3"], "split_rhat_max": _sig(v["split_rhat_max"], 5),
                            "n_divergent": v["n_divergent"], "run": v["run"]}
    if "full_document" in full:
        line["full_document"] = full["full_document"]
    legs = {}
    if "mh" in full:
        legs["mh"] = _leg(full["mh"], ("accept_rate",))
        legs["mh"]["workload"] = "adaptive_mcmc_chain reference_model(20), 200 adapting + 400 sampling steps, all timed"
        pr = full["mh"].get("per_gpu_kernel_rate") or {}
        if pr:
            legs["mh"]["adapting"], legs["mh"]["sampling"] = _sig(pr["adapting"], 3), _sig(pr["sampling"], 3)
        for k in ("chains_8192",):
            if k in full["mh"]:
                legs["mh_8192"] = _leg(full["mh"][k])
    if "c5" in full:
        legs["c5"] = _leg(full["c5"])
        legs["c5"]["workload"] = "C5 mixture S=68 O=64, 262144 chains (weak) / sharded (strong), sampling"
        if "chains_32768" in full["c5"]:
            legs["c5_32768"] = _leg(full["c5"]["chains_32768"])
    if "smc" in full:
        legs["smc"] = _leg(full["smc"], ("seconds_per_run", "tempering_steps", "log_evidence"))
        legs["smc"]["adaptation"] = "gpu batched / cpu sequential (reference order)"
    if "c3" in full:
        for tag in ("chains_65536", "chains_8192"):
            if tag in full["c3"]:
                e = dict(full["c3"][tag]); e["unit"] = full["c3"]["unit"]
                if tag == "chains_65536" and "cpu_baseline" in full["c3"]:
                    e["cpu_baseline"] = full["c3"]["cpu_baseline"]
                legs["c3_" + tag[7:]] = _leg(e)
    if "hmc_fd_dense" in full:
        legs["hmc_fd_dense"] = _leg(full["hmc_fd_dense"])
        legs["hmc_fd_dense"]["workload"] = "headline model, grad_log_joint verbatim (the reference's arithmetic)"
    if "hmc_8192" in full:
        legs["hmc_8192"] = _leg(full["hmc_8192"])
    if legs:
        line["legs"] = legs
    return line


def emit(args, full):
    """Rank 0: the verbose document to --full-out (best effort), the compact line (or, with --full, the document) to stdout."""
    if args.full_out:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(args.full_out)), exist_ok=True)
            with open(args.full_out, "w") as f:
                json.dump(full, f)
            full["full_document"] = args.full_out
        except OSError as ex:
            full["full_document"] = f"not written ({ex})"
    print(json.dumps(full if args.full else compact(full), separators=(",", ":")), flush=True)


def timed_hmc(X, cp, C, cfg, n_warm, n_timed, per_launch, repeats, seed=1, with_draws=False, local=False):
    """`repeats` timed regions of `n_timed` transitions on one engine (after `n_warm` untimed ones); returns (list of
    region seconds, avg full-launch ms, transitions per launch, kernel, stats).  local: a leg that rank 0 runs alone -- its
    clock must not enter a collective."""
    E, torch, stream = X.E, X.torch, X.stream
    clock = Clock(torch, X.dist, 1, X.coll_dev) if local else X.clock
    eng = E.Engine(cp, C, seed=seed, chain_offset=X.rank * C, device=X.dev)
    eng.set_stream(stream.cuda_stream)
    eng.hmc_init(cfg, 0)
    d_draws = eng.device_alloc(max(1, n_timed) * cp.d * C * 8) if with_draws else None
    if n_warm > 0:
        eng.hmc_step(n_warm)
    dts, events = [], []
    for _ in range(repeats):
        ev = []
        dts.append(clock.region(lambda: ev.extend(stepped(torch, stream, lambda n, done: eng.hmc_step(n, (d_draws + done * cp.d * C * 8) if d_draws else None), n_timed, per_launch))))
        events.extend(ev)
    launch_ms, n_launch = full_launch_ms(events)
    kernel, st = eng.hmc_last_kernel(), eng.hmc_stats()
    if d_draws:
        eng.device_free(d_draws)
    eng.close()
    return dts, launch_ms, n_launch, kernel, st


def leg_c3(X):
    """BASELINE configs[2]: linear_regression.rs ridge form, 32 Normal coefficients x 1 024 synthetic observations
    (examples/linear_regression.rs:396-424 generalised) -- the coordinates interact through every observation.
    Two chain counts: 65 536 per GPU, and 8 192 per GPU = BASELINE's own 8-GPU sharding of a 65 536-chain job."""
    E, W, world = X.E, X.W, X.world
    Xd, y, _ = W.ridge_data(C3_N, C3_P)
    cp = E.compile_model(W.ridge_regression(Xd, y))
    L, d = 16, cp.d
    cfg = E.hmc_config(n_leapfrog=L, init_step_size=0.004)
    # SURVEY 8d, FD reference semantics: per gradient 2 d whole-model runs of S prior log-pdfs + O x (1 log-pdf + d multiply-adds)
    flops_grad = 2 * d * (cp.S * FLOPS_PER_NORMAL_LOGPDF + cp.O * (FLOPS_PER_NORMAL_LOGPDF + 2 * d))
    flops_step = flops_grad * (L + 1) / L
    out = {"metric": "hmc_leapfrog_steps_per_sec", "unit": "leapfrog-steps/s", "n_gpus": world,
           "config": {"workload": f"C3: hmc_chain, ridge regression beta#j~N(0,1), y#i~N(sum_j beta#j X[i][j], 0.5), d={d}, O={cp.O}, L=16, fixed step 0.004, "
                                  "fd_sparse (engine default)", "grad": "fd_sparse"},
           "flops_per_leapfrog_step_dense_semantics": flops_step,
           "flops_note": "SURVEY 8d: 2 d (S x 8 + O x (8 + 2 d)) flops per gradient x (L+1)/L -- the reference's 2 d whole-model runs; the kernel shares the "
                         "products and prefix sums between coordinates and executes fewer (see roofline.executed)"}
    for tag, total in (("chains_65536", 65536), ("chains_8192", 8192)):
        C = shard(total, world, X.args.scaling) if X.args.scaling == "strong" else total
        nt = 3 if total == 65536 else 6
        dts, launch_ms, n_launch, kernel, st = timed_hmc(X, cp, C, cfg, 1, nt, 1, 3, seed=3)
        dt = float(np.median(dts))
        val = world * C * nt * L / dt
        ent, src = pmc_entry(f"hmc|c3|{C}|fd_sparse|L{L}")
        tfl = C * n_launch * flops_step * L / (launch_ms * 1e-3) / 1e12
        # what k_hmc_lin_steps EXECUTES per (chain, observation, gradient): every wave forms the D products and the D prefix additions,
        # carries the two suffix chains of its M = D / W coordinates (2 (D - p) additions at position p) and two densities per
        # coordinate (8 operations each); half tiles put the two signs into the two lane halves (both halves form products and
        # prefixes).  One flop per add / mul (unfused: the reference's rounding).  DESIGN 3.9; tools/isa_loops.py counts the same loop.
        import re as _re
        Wl = int((_re.search(r"W=(\d+)", kernel) or [0, max(1, d // 4)])[1])
        Ml = max(1, d // max(1, Wl))
        half = "half" in kernel
        per_obs = (Wl * (4 * d + 16 * Ml) + d * (d + 1)) if half else (Wl * (2 * d + 16 * Ml) + d * (d + 1))
        exec_grad = cp.O * per_obs + 2 * d * FLOPS_PER_NORMAL_LOGPDF          # + each coordinate's prior record at q_j +- h
        exe = C * n_launch * exec_grad * (L + 1) / (launch_ms * 1e-3) / 1e12 if "k_hmc_lin" in kernel else None
        pm = executed_from_pmc(ent, n_launch, launch_ms * 1e-3)
        frac_exec = (exe if exe is not None else (pm or {}).get("f64_tflops", tfl)) / F64_VALU_PEAK_TFLOPS
        out[tag] = {"value": val, "chains_per_gpu": C, "seconds_per_transition": dt / nt, "timed_regions": {"repeats": len(dts), "steps_each": nt, "value": spread([world * C * nt * L / t for t in dts])},
                    "accept_rate": st.accept_rate,
                    "roofline": {"bound": "valu_f64", "achieved": frac_exec * F64_VALU_PEAK_TFLOPS, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": frac_exec,
                                 "frac_kind": "executed f64 add / mul (1 flop each) of the launch / HIP-event time",
                                 "reference_implied": {"achieved": tfl, "frac": tfl / F64_VALU_PEAK_TFLOPS,
                                                       "note": "SURVEY 8d's dense-semantics flops (the reference's 2 d whole-model runs per gradient) at this rate: "
                                                               "NOT executed -- the kernel shares products and prefix sums between coordinates"},
                                 "kernel": kernel, "avg_launch_ms": launch_ms, "traffic": traffic_from_pmc(ent, n_launch), "traffic_source": src,
                                 "executed": pm, "executed_source": src,
                                 "note": "achieved / frac = the f64 operations the kernel executes (counted from its loop structure: per observation every wave forms D "
                                         "products + D prefix additions + the suffix chains and densities of its own coordinates) over HIP-event time; unfused "
                                         "add / mul tops out at half of the FMA peak; `executed` = the same from the committed PMC entry when this configuration has one"}}
    return out


def leg_hmc_small(X, C):
    """The headline model at the chain count a GPU holds when BASELINE's 65 536-chain job is sharded 8 x (strong scaling): the latency
    regime (half tiles).  Rank 0 alone; same config as the headline."""
    E, W = X.E, X.W
    cp = E.compile_model(W.normal_sites(N_SITES))
    L, d, nt = 16, cp.d, 200
    dts, launch_ms, n_launch, kernel, st = timed_hmc(X, cp, C, E.hmc_config(n_leapfrog=L), 100, nt, 25, 3, local=True)
    dt = float(np.median(dts))
    evals = (2 * d * (L + 1)) * 2 + 2 * N_SITES
    tfl = C * n_launch * evals * FLOPS_PER_NORMAL_LOGPDF / (launch_ms * 1e-3) / 1e12
    ent, src = pmc_entry(f"hmc|normal32|{C}|fd_sparse|L{L}")
    return {"metric": "hmc_leapfrog_steps_per_sec", "value": C * nt * L / dt, "unit": "leapfrog-steps/s", "n_gpus": 1,
            "timed_regions": {"repeats": len(dts), "steps_each": nt, "value": spread([C * nt * L / t for t in dts])},
            "config": {"workload": f"C2-normal32, {C} chains, L=16, fd_sparse, 100 adaptive warmup transitions untimed"},
            "roofline": {"bound": "valu_f64", "achieved": tfl, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tfl / F64_VALU_PEAK_TFLOPS,
                         "kernel": kernel, "avg_launch_ms": launch_ms, "traffic": traffic_from_pmc(ent, n_launch), "traffic_source": src,
                         "executed": executed_from_pmc(ent, n_launch, launch_ms * 1e-3), "executed_source": src}}


def leg_dense(X):
    """The reference's arithmetic verbatim on the headline model (FG_GRAD_FD_DENSE: every g_i is the difference of two WHOLE
    log-joints, hmc.rs:304-329) -- the only GPU figure the CPU baseline may be divided into."""
    E, W = X.E, X.W
    C, L = X.args.chains if X.args.scaling == "weak" else shard(X.args.chains, X.world, "strong"), 16
    cp = E.compile_model(W.normal_sites(N_SITES))
    nt = 50
    dts, launch_ms, n_launch, kernel, st = timed_hmc(X, cp, C, E.hmc_config(grad_mode=E.GRAD_FD_DENSE), 10, nt, 25, 3, local=True)
    dt = float(np.median(dts))
    n_stmt, d = 2 * N_SITES, cp.d
    evals_sem = (2 * d * (L + 1)) * n_stmt + n_stmt
    evals_exec = (L + 1) * (n_stmt + 2 * 2 * d) + n_stmt
    # the in-order additions the kernel performs per gradient: both whole-run sums of a coordinate are the same number ahead of its
    # own row (that prefix is added once per wave and extended from one own coordinate to the next), two chains behind it
    import re as _re
    Wd = int((_re.search(r"W=(\d+)", kernel) or [0, 16])[1])
    n_w = max(1, d // max(1, Wd))
    prefix = n_w * Wd * (Wd - 1) // 2 + Wd * (n_w - 1)
    adds = (L + 1) * 2 * (d * (d - 1) + 2 * d + prefix)                       # x 2: log_prior and log_likelihood (one statement of each per coordinate)
    adds_sem = (L + 1) * 2 * d * n_stmt                                       # ... of the 2 d whole scoring runs as the reference performs them
    sem = C * n_launch * evals_sem * FLOPS_PER_NORMAL_LOGPDF / (launch_ms * 1e-3) / 1e12
    exe = C * n_launch * (evals_exec * FLOPS_PER_NORMAL_LOGPDF + adds) / (launch_ms * 1e-3) / 1e12
    ent, src = pmc_entry(f"hmc|normal32|{C}|fd_dense|L{L}")
    return {"metric": "hmc_leapfrog_steps_per_sec", "value": C * nt * L / dt, "unit": "leapfrog-steps/s", "n_gpus": 1,
            "timed_regions": {"repeats": len(dts), "steps_each": nt, "value": spread([C * nt * L / t for t in dts])},
            "config": {"workload": f"C2-normal32, {C} chains, L=16, FG_GRAD_FD_DENSE = grad_log_joint verbatim", "grad": "fd_dense"},
            "roofline": {"bound": "valu_f64", "achieved": exe, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": exe / F64_VALU_PEAK_TFLOPS,
                         "kernel": kernel, "avg_launch_ms": launch_ms, "traffic": traffic_from_pmc(ent, n_launch), "traffic_source": src,
                         "executed": executed_from_pmc(ent, n_launch, launch_ms * 1e-3), "executed_source": src,
                         "dense_semantics_tflops": sem,
                         "in_order_additions_per_gradient": {"performed": adds // (L + 1), "of_the_2d_whole_runs": adds_sem // (L + 1)},
                         "note": "achieved = what the dense kernel computes: every statement's density once per gradient + the moved ones at +-h (x 8 flops) + the "
                                 "in-order additions it performs (1 flop each: the part of the two whole-run sums ahead of a coordinate's own row is one number, "
                                 "added once per wave; the same bits as 2 d whole scoring runs); dense_semantics_tflops = SURVEY 8d's 2 d (S+O) log-pdfs x 8 flops per "
                                 "gradient at this rate (re-evaluating densities that did not move), which exceeds the peak and is NOT executed"}}


def leg_mh(X):
    """The MCMC half of BASELINE's metric: adaptive_mcmc_chain on the reference's own bench model
    (benches/f_perf.rs:78-109: reference_model(20), 20 sample + 19 observe sites) at 65 536 chains per GPU; chain steps are
    counted over warmup + sampling: 200 adapting + 400 sampling steps, all timed.  `chains_8192`: the same at the
    8 192 chains a GPU holds when a 65 536-chain job is sharded 8 x (the latency regime)."""
    args, world = X.args, X.world
    C = shard(args.chains, world, args.scaling)
    out = _mh_refmodel(X, C, args.spinup)
    if args.scaling == "weak" and C != 8192:
        out["chains_8192"] = _mh_refmodel(X, 8192, 0.0)
    return out


def _mh_refmodel(X, C, spinup):
    E, W, torch, clock, stream, world, rank, dev = X.E, X.W, X.torch, X.clock, X.stream, X.world, X.rank, X.dev
    cp = E.compile_model(W.reference_model(20))
    eng = E.Engine(cp, C, seed=1, chain_offset=rank * C, device=dev)
    eng.set_stream(stream.cuda_stream)
    nw, ns, per = 200, 400, 100
    eng.mh_init(nw)
    eng.mh_step(per)                                       # untimed: first-launch effects
    scratch = None
    if spinup > 0:                                         # clock spin-up on a scratch engine, as in the HMC leg
        scratch = E.Engine(cp, C, seed=987654321, chain_offset=rank * C, device=dev)
        scratch.set_stream(stream.cuda_stream)
        scratch.mh_init(0)
        t_sp = time.perf_counter()
        while time.perf_counter() - t_sp < spinup:
            scratch.mh_step(4 * per)
            torch.cuda.synchronize()
    dts, events, phase = [], [], {"adapting": [], "sampling": []}
    for _ in range(3):
        eng.mh_init(nw)
        ev = []
        dts.append(clock.region(lambda: ev.extend(stepped(torch, stream, lambda n, done: eng.mh_step(n), nw + ns, per))))
        events.extend(ev)
        ms = [e0.elapsed_time(e1) for e0, e1, _ in ev]
        phase["adapting"].append(C * nw / (sum(ms[: nw // per]) * 1e-3))
        phase["sampling"].append(C * ns / (sum(ms[nw // per:]) * 1e-3))
    launch_ms, n_launch = full_launch_ms(events)
    if scratch is not None:
        scratch.close()
    acc = eng.mh_stats().accept_rate
    mh_kernel = eng.mh_last_kernel()
    eng.close()
    dt = float(np.median(dts))
    S, O = cp.S, cp.O
    bytes_per_step = 8 * S + 40                            # value row + 1 value + adaptation RMW + lw
    tflops = C * n_launch * (S + O) * FLOPS_PER_NORMAL_LOGPDF / (launch_ms * 1e-3) / 1e12
    ent, src = pmc_entry(f"mh|refmodel20|{C}")
    return {"metric": "mh_chain_steps_per_sec", "value": world * C * (nw + ns) / dt, "unit": "chain-steps/s", "n_gpus": world,
            "accept_rate": acc, "timed_regions": {"repeats": len(dts), "steps_each": nw + ns, "value": spread([world * C * (nw + ns) / t for t in dts])},
            "per_gpu_kernel_rate": {"adapting": float(np.median(phase["adapting"])), "sampling": float(np.median(phase["sampling"])),
                                    "note": "chain-steps/s of this rank's launches by HIP events, the two phases apart"},
            "config": {"workload": f"adaptive_mcmc_chain, reference_model(20) (benches/f_perf.rs:78-91: S=20, O=19), {C} chains/GPU, "
                                   f"{nw} adapting + {ns} sampling steps, all timed", "steps_per_launch": n_launch},
            "roofline": {"bound": "valu_f64", "achieved": tflops, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / F64_VALU_PEAK_TFLOPS,
                         "traffic": traffic_from_pmc(ent, n_launch), "traffic_source": src, "kernel": mh_kernel, "avg_launch_ms": launch_ms,
                         "executed": executed_from_pmc(ent, n_launch, launch_ms * 1e-3),
                         "note": "(S + O) log-pdfs x 8 flops per chain step / HIP-event time",
                         "hbm_nominal": {"achieved": C * n_launch * bytes_per_step / (launch_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "note": "SURVEY 8d: 8 S + 40 B per chain step; the value row lives in LDS across a launch"}},
            "published_reference": "65 k chain-steps/s/thread (15.3 us per transition, Apple Silicon; benches/f_perf.rs:24-28)"}


def leg_c5(X):
    """BASELINE configs[4]: 4-component Gaussian mixture (4 f64 + 64 usize sites, 64 observations), adaptive_mcmc_chain at
    262 144 chains (per GPU in the weak mode; in total, sharded, in the strong mode = BASELINE's 8 x 32 768).  `chains_32768`:
    BASELINE's per-GPU share, measured on this GPU."""
    args, world = X.args, X.world
    C = 262144 if args.scaling == "weak" else shard(262144, world, "strong")
    out = _mh_c5(X, C)
    if args.scaling == "weak":
        out["chains_32768"] = _mh_c5(X, 32768)
    return out


def _mh_c5(X, C):
    E, W, torch, clock, stream, world, rank, dev = X.E, X.W, X.torch, X.clock, X.stream, X.world, X.rank, X.dev
    data, _ = W.mixture_data(64)
    cp = E.compile_model(W.mixture(data))
    eng = E.Engine(cp, C, seed=1, chain_offset=rank * C, device=dev)
    eng.set_stream(stream.cuda_stream)
    eng.mh_init(200)
    eng.mh_step(200)
    dts, events = [], []
    for _ in range(3):
        ev = []
        dts.append(clock.region(lambda: ev.extend(stepped(torch, stream, lambda n, done: eng.mh_step(n), 200, 100))))
        events.extend(ev)
    launch_ms, n_launch = full_launch_ms(events)
    mh_kernel = eng.mh_last_kernel()
    eng.close()
    dt = float(np.median(dts))
    tflops = C * n_launch * (cp.S + cp.O) * FLOPS_PER_NORMAL_LOGPDF / (launch_ms * 1e-3) / 1e12
    ent, src = pmc_entry(f"mh|c5|{C}")
    return {"metric": "mh_chain_steps_per_sec", "value": world * C * 200 / dt, "unit": "chain-steps/s", "n_gpus": world,
            "timed_regions": {"repeats": len(dts), "steps_each": 200, "value": spread([world * C * 200 / t for t in dts])},
            "config": {"workload": f"C5: 4-component mixture, S={cp.S} (4 f64 + 64 usize), O={cp.O}, {C} chains/GPU, 200 sampling steps after 200 adapting"},
            "roofline": {"bound": "valu_f64", "achieved": tflops, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / F64_VALU_PEAK_TFLOPS,
                         "traffic": traffic_from_pmc(ent, n_launch), "traffic_source": src, "kernel": mh_kernel, "avg_launch_ms": launch_ms,
                         "executed": executed_from_pmc(ent, n_launch, launch_ms * 1e-3),
                         "note": "(S + O) log-pdfs x 8 flops per chain step / HIP-event time; Categorical table lookups counted as log-pdfs",
                         "hbm_nominal": {"achieved": C * n_launch * (8 * cp.S + 40) / (launch_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s"}}}


def leg_smc(X):
    """C4: adaptive_smc, 1 048 576 particles, Systematic / ESS 0.5 / 3 rejuvenation moves (examples/smc_inference.rs:36-65).
    SMC does not shard without communication (next_beta, normalisation and resampling are global): N GPUs = N independent
    replicas with different seeds -- no collective is invented."""
    args, E, W, clock, stream, world, rank, dev = X.args, X.E, X.W, X.clock, X.stream, X.world, X.rank, X.dev
    N = SMC_PARTICLES
    cp = E.compile_model(W.smc_normal())
    eng = E.Engine(cp, N, seed=42 + rank, device=dev)
    eng.set_stream(stream.cuda_stream)
    eng.smc_run(rejuvenation_steps=3, download=False)      # untimed: allocations, first-launch effects
    scratch = None
    if args.spinup > 0:                                    # clock spin-up: the same run on a scratch population
        scratch = E.Engine(cp, N, seed=987654321 + rank, device=dev)
        scratch.set_stream(stream.cuda_stream)
        t_sp = time.perf_counter()
        while time.perf_counter() - t_sp < args.spinup:
            scratch.smc_run(rejuvenation_steps=3, download=False)
    res, dts = {}, []
    for _ in range(5):
        dts.append(clock.region(lambda: res.update(eng.smc_run(rejuvenation_steps=3, download=False))))   # particles and weights stay in HBM
    if scratch is not None:
        scratch.close()
    eng.close()
    dt = float(np.median(dts))
    n_steps = len(res["betas"])
    moves = (res["n_model_runs"] - N) / 2
    S = cp.S
    per_particle_step = 24 + 1040 + 12 + 16 * S + 3 * (16 * S + 24)      # reweight + next_beta (65 passes x 16 B) + resample + gather + rejuvenation
    gbs = N * n_steps * per_particle_step / dt / 1e9
    ent, src = pmc_entry(f"smc|c4|{N}")
    return {"metric": "smc_particle_moves_per_sec", "value": world * moves / dt, "unit": "particle-moves/s", "n_gpus": world,
            "seconds_per_run": dt, "seconds_per_run_spread": spread(dts), "tempering_steps": n_steps, "log_evidence": res["log_evidence"],
            "log_evidence_closed_form": -1.9305103088617774,
            "config": {"workload": f"C4: adaptive_smc, {N} particles, Systematic / 0.5 / 3 rejuvenation moves, mu~N(0,1); y~N(mu,0.5)=1.5",
                       "adaptation": "batched: the shared DiminishingAdaptation is updated once per rejuvenation sweep from per-site counts (DESIGN deviation ii); the "
                                     "CPU baseline beside it runs the reference's sequential form (one update per particle)",
                       "sharding": "replicas only" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                         "frac_kind": "ALGORITHMIC bytes (SURVEY 8d) / host wall time of the run -- not counter traffic",
                         "counter_traffic_frac": (traffic_from_pmc(ent, 1) / dt / 1e9 / HBM_PEAK_GBS) if traffic_from_pmc(ent, 1) else None,
                         "traffic": traffic_from_pmc(ent, 1), "traffic_source": src,
                         "kernel": "whole fg_smc_run (next_beta bisection + reweight + scan + resample + gather + rejuvenation)",
                         "bytes_per_particle_per_tempering_step": per_particle_step,
                         "note": "SURVEY 8d algorithmic bytes x particles x tempering steps / wall time of the whole run (host-timed, launch gaps "
                                 "included); the 16 MB of (ll, lw) fit the L2 / Infinity Cache, so HBM is not what bounds the 65 ESS evaluations"}}


def extras(X):
    """Side measurements on rank 0: the opt-in analytic gradient and C2 as BASELINE.json words it (the README model)."""
    E, W, dev = X.E, X.W, X.dev
    out = {}
    C = X.args.chains if X.args.scaling == "weak" else shard(X.args.chains, X.world, "strong")
    cp = E.compile_model(W.normal_sites(N_SITES))
    eng = E.Engine(cp, C, seed=1, device=dev)
    eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_ANALYTIC), 0)
    eng.hmc_step(25); eng.synchronize()
    t0 = time.perf_counter(); eng.hmc_step(100); eng.synchronize(); dt = time.perf_counter() - t0
    out["hmc_analytic_leapfrog_steps_per_sec"] = C * 100 * 16 / dt
    out["hmc_analytic_note"] = "closed-form derivative of the Normal force terms: NOT the reference's arithmetic, opt-in, never `value`"
    eng.close()
    progress(X.rank, "extras: analytic gradient done")
    # a model the record streams do not cover (expression parameters -> the interpreter kernels): the reference's own logistic
    # regression example (examples/classification.rs:104-135: 100 observations, 3 coefficients, run there with adaptive_mcmc_chain)
    Xc, yc, _ = W.classification_data(100)
    cpl = E.compile_model(W.logistic_regression(Xc, yc))
    for Cl in (C, 8192):
        try:                                                # a side measurement must not cost the run its line
            eng = E.Engine(cpl, Cl, seed=1, device=dev)
            eng.mh_init(100); eng.mh_step(200); eng.synchronize()
            t0 = time.perf_counter(); eng.mh_step(400); eng.synchronize(); dt = time.perf_counter() - t0
            out[f"interpreter_logistic_regression_mh_chain_steps_per_sec_{Cl}_chains"] = Cl * 400 / dt
            eng.hmc_init(E.hmc_config(), 5); eng.hmc_step(5); eng.synchronize()
            t0 = time.perf_counter(); eng.hmc_step(10); eng.synchronize(); dt = time.perf_counter() - t0
            out[f"interpreter_logistic_regression_hmc_leapfrog_steps_per_sec_{Cl}_chains"] = Cl * 10 * 16 / dt
            out[f"interpreter_kernels_{Cl}_chains"] = [eng.mh_last_kernel(), eng.hmc_last_kernel()]
            eng.close()
        except E.EngineError as ex:
            out[f"interpreter_error_{Cl}_chains"] = str(ex)
        progress(X.rank, f"extras: logistic regression at {Cl} chains done")
    out["interpreter_note"] = ("examples/classification.rs logistic regression (S = 3, O = 100; prob = clamp(1 / (1 + exp(-x.beta)))): no record stream -- the "
                               "program is compiled at run time (k_mh_jit_steps / k_hmc_jit_steps; hiprtc, cached on disk), or runs on the multi-wave "
                               "interpreter kernels when hiprtc is absent; side measurement, never `value`")
    # C2 as BASELINE.json words it: the README model (d = 1), 65 536 chains x 1 000 steps after 200 warmup transitions
    cp1 = E.compile_model(W.readme_normal())
    eng = E.Engine(cp1, CHAINS_PER_GPU, seed=1, device=dev)
    d1 = eng.device_alloc(1000 * cp1.d * CHAINS_PER_GPU * 8)
    eng.hmc_run(E.hmc_config(), 50, 50, d1); eng.synchronize()
    eng.close()
    eng = E.Engine(cp1, CHAINS_PER_GPU, seed=1, device=dev)
    t0 = time.perf_counter(); eng.hmc_run(E.hmc_config(), 1000, 200, d1); eng.synchronize(); dt = time.perf_counter() - t0
    out["hmc_readme_model_leapfrog_steps_per_sec"] = CHAINS_PER_GPU * 1200 * 16 / dt
    out["hmc_readme_model_workload"] = "C2 README model (mu~N(0,1); y~N(mu,0.5)=1.2), 65536 chains, 200 warmup + 1000 sampling transitions, L=16: %.1f ms" % (dt * 1e3)
    eng.device_free(d1)
    eng.close()
    return out


def validity(E, W, D, dev):
    """The north_star acceptance test at a fixed length, whatever --steps / --warmup are: 65 536 chains, 200 warmup + 200
    sampling transitions of the headline model in the default gradient mode."""
    C, nw, ns = CHAINS_PER_GPU, 200, 200
    cp = E.compile_model(W.normal_sites(N_SITES))
    eng = E.Engine(cp, C, seed=2, device=dev)
    d_draws = eng.device_alloc(ns * cp.d * C * 8)
    st = eng.hmc_run(E.hmc_config(grad_mode=E.GRAD_FD_SPARSE), ns, nw, d_draws)
    r = eng.diag_rhat_ess(d_draws, ns, cp.d, None)          # this engine's chains only: rank 0 runs this leg alone, so NO collective here
    rhat, mean = r["r_hat"], r["mean"]
    eng.device_free(d_draws)
    eng.close()
    _, tm, _ = W.normal_sites_truth(N_SITES)
    err = float(np.abs(mean - tm).max())
    return {"run": f"{C} chains, {nw} warmup + {ns} sampling transitions, fd_sparse, L=16", "posterior_mean_max_abs_err": err,
            "target_1e-3": bool(err <= 1e-3), "split_rhat_max": float(np.max(rhat)), "split_rhat_lt_1.01": bool(np.max(rhat) < 1.01),
            "accept_rate": st.accept_rate, "n_divergent": int(st.n_divergent)}


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rc = run_rank(args)
    if ABANDONED or rc:                                          # a thread may be stuck inside a collective: do not wait for it at interpreter exit
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(rc or EXIT_DIAGNOSTICS_FAILED)


if __name__ == "__main__":
    main()
