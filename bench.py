#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: leapfrog-steps/s (HMC) + chain-steps/s (MCMC) at 65 536
chains per GPU, with SMC beside them.

Headline workload (BASELINE.json configs[1] + north_star): `hmc_chain` on the 32-site conjugate Normal model
(x#i ~ N(0,1); y#i ~ N(x#i, 0.5) observed at 0.2 i - 1), 65 536 chains per GPU, HMCConfig::default() (L = 16, h = 1e-5,
target accept 0.8), gradient mode FG_GRAD_FD_SPARSE -- the engine's ONE default (C ABI, Python and this file): the
reference's central difference (hmc.rs:304-329) over the statements that read the perturbed coordinate.  The
reference-verbatim dense mode (2 d whole-model runs per gradient) is measured beside it (`hmc_fd_dense`) and is what the
CPU baseline is compared with like for like.

A bench "step" = ONE HMC transition (16 leapfrog steps, 17 gradients, endpoint score, accept/reject, adaptation) of EVERY
chain.  `--warmup W` untimed transitions are the adaptive warmup (timed separately and reported as
`incl_warmup`), the `--steps K` timed ones are sampling transitions whose draws go to a [K][d][C] buffer in HBM -- what
`hmc_chain` returns.  value = world x chains x K x L / time (weak scaling: every rank runs its own 65 536 chains, Philox
keyed by the global chain id; no data-path collective; the R-hat all-gather runs after the timed region).

`--gpus N` without a torchrun environment starts the N ranks itself (a `python -m torch.distributed.run` child, before
anything here touches HIP) and relays rank 0's line.

One JSON line on stdout (rank 0):
  roofline      the dominant kernel (k_hmc_sep_steps): algorithmic log-pdf evaluations x 8 flops / HIP-event time against the f64
                vector peak; the SURVEY 8d HBM-nominal figure (state as if it round-tripped HBM) is kept as a note only --
                the kernel keeps q, p in registers; `traffic` = measured FETCH_SIZE + WRITE_SIZE per launch (profiles/).
  cpu_baseline  the CPU oracle (C restatement of the reference algorithm, dense FD) on this box's host cores, bounded
                sample, beside the dense-FD GPU rate.
  mh, smc       the other halves of the metric: adaptive_mcmc_chain on the reference's own bench model
                (benches/f_perf.rs:78-109) and adaptive_smc at 1 048 576 particles, each with its own roofline and
                cpu_baseline (single-thread calibration against the published 15.3 / 73.1 us per transition).
  validity      a fixed 200 + 200 run of the headline model at 65 536 chains (independent of --steps/--warmup):
                |pooled mean - closed form| <= 1e-3 (north_star), split R-hat.
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
F64_VALU_PEAK_TFLOPS = 78.6    # 256 CU x 4 SIMD x 32 lanes x 2 flop (FMA) x 2.4 GHz / 2 (f64 half rate)
FLOPS_PER_NORMAL_LOGPDF = 8.0  # a Normal log-pdf with ln(sigma) hoisted ~ 8 flops
SMC_PARTICLES = 1 << 20


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--chains", type=int, default=CHAINS_PER_GPU, help="chains per GPU")
    ap.add_argument("--grad", choices=["fd_sparse", "fd_dense", "analytic"], default="fd_sparse",
                    help="fd_sparse (engine default) / fd_dense (reference verbatim): the reference's central difference; "
                         "analytic: closed-form derivative (not the reference's arithmetic)")
    ap.add_argument("--launch", type=int, default=25, help="transitions fused per kernel launch")
    ap.add_argument("--leapfrog", type=int, default=16, help="L (HMCConfig::default is 16; other values are for experiments only)")
    ap.add_argument("--spinup", type=float, default=0.4, help="seconds of untimed throw-away transitions on a scratch engine before the measured "
                    "engine starts (the GPU's clocks settle over the first tens of ms of f64 load); 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the MH / SMC / dense-FD / validity legs")
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
                                "note": "the oracle has only the reference's dense FD; compare it with the GPU's dense mode, not with the sparse headline"}
    return out


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
            "sample": f"adaptive_smc, {n} particles, Systematic / 0.5 / 3 rejuvenation moves, {len(r['betas'])} tempering steps, {dt:.2f} s wall; "
                      "sequential by construction (one shared DiminishingAdaptation, smc.rs:482); C restatement, not the Rust binary",
            "seconds_scaled_to_1048576_particles": dt * (SMC_PARTICLES / n)}


# ------------------------------------------------------------------------------------------ profile lookups
def measured_traffic(chains, n_launch, grad):
    """HBM bytes per launch of the HMC kernel from the committed rocprofv3 PMC passes (FETCH_SIZE + WRITE_SIZE in separate
    --pmc passes, profiles/*hbm_traffic.json), scaled by transitions per launch; null when the run is not the profiled
    configuration (chain count, gradient mode)."""
    for name in ("round2_hbm_traffic.json", "round1_hbm_traffic.json"):
        try:
            p = json.load(open(os.path.join(ROOT, "profiles", name)))
            c = p["config"]
            if (c["chains"], c["grad"]) == (chains, grad):
                return p["sampling_launch_bytes"]["total"] / c["transitions_per_launch"] * n_launch
        except Exception:
            continue
    return None


def measured_traffic_mh(chains, n_adapting, n_sampling, steps_per_launch):
    """HBM bytes per launch of the MH kernel on reference_model(20), averaged over the leg's launches, from the committed PMC
    passes (bytes per chain step of an adapting / a sampling launch); null when the profile is absent."""
    try:
        b = json.load(open(os.path.join(ROOT, "profiles", "round2_hbm_traffic.json")))["mh"]["reference_model20_bytes_per_chain_step"]
        per_step = (b["adapting"] * n_adapting + b["sampling"] * n_sampling) / (n_adapting + n_sampling)
        return per_step * chains * steps_per_launch
    except Exception:                                                          # noqa: BLE001
        return None


def measured_traffic_smc():
    """HBM bytes of one adaptive_smc run at 1 048 576 particles (FETCH_SIZE + WRITE_SIZE summed over its kernels)."""
    try:
        s = json.load(open(os.path.join(ROOT, "profiles", "round2_hbm_traffic.json")))["smc"]
        return s["fetch_bytes_per_run"] + s["write_bytes_per_run"]
    except Exception:                                                          # noqa: BLE001
        return None


NATIVE_RCCL_TIMEOUT_S = float(os.environ.get("FG_BENCH_RCCL_TIMEOUT", "120"))


def progress(rank, msg):
    """One stderr line per finished leg (stdout carries the single JSON line)."""
    sys.stderr.write(f"[bench rank {rank} +{time.perf_counter() - T_START:.1f}s] {msg}\n")
    sys.stderr.flush()


T_START = time.perf_counter()


def call_with_timeout(fn, seconds):
    """fn() on a daemon thread (ctypes calls release the GIL); raises TimeoutError when it has not returned in time -- the
    thread is then abandoned (the process ends with os._exit in that case, see main)."""
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


def executed_from_pmc(chains, n_launch, grad, launch_ms):
    """What the dominant kernel actually issues per launch, from the committed PMC passes (profiles/round2_hmc_pmc.json) and
    this run's launch time: executed f64 flops (add / mul = 1, fma = 2 per lane) and the share of SIMD cycles that issue a
    VALU instruction (every wave64 VALU instruction holds its SIMD for 4 cycles).  null when the run is not the profiled
    configuration."""
    try:
        p = json.load(open(os.path.join(ROOT, "profiles", "round2_hmc_pmc.json")))
        c, m = p["config"], p["per_launch"]
        if (c["chains"], c["grad"]) != (chains, grad) or n_launch < 1:
            return None
        k = n_launch / c["transitions_per_launch"]                               # counts scale with the transitions of a launch
        m = {key: v * k for key, v in m.items()}
        flops = 64.0 * (m["SQ_INSTS_VALU_ADD_F64"] + m["SQ_INSTS_VALU_MUL_F64"] + 2.0 * m["SQ_INSTS_VALU_FMA_F64"])
        f64 = m["SQ_INSTS_VALU_ADD_F64"] + m["SQ_INSTS_VALU_MUL_F64"] + m["SQ_INSTS_VALU_FMA_F64"]
        simd_cycles = launch_ms * 1e-3 * 2.1e9 * 1024                          # 256 CUs x 4 SIMDs at the ~2.1 GHz the clock holds under f64 load
        return {"f64_tflops": flops / (launch_ms * 1e-3) / 1e12, "f64_share_of_valu": f64 / m["SQ_INSTS_VALU"],
                "valu_issue_share_of_simd_cycles_at_2.1GHz": 4.0 * m["SQ_INSTS_VALU"] / simd_cycles,
                "note": "instruction counts from profiles/round2_hmc_pmc.json (rocprofv3 --pmc, same configuration) over this run's launch time"}
    except Exception:                                                          # noqa: BLE001
        return None


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
    torch_sync_needed = [(e0.elapsed_time(e1), n) for e0, e1, n in events]
    n0 = events[0][2]
    return float(np.mean([ms for ms, n in torch_sync_needed if n == n0])), n0


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
    C, K, Wn, L = args.chains, args.steps, args.warmup, args.leapfrog
    cp = E.compile_model(W.normal_sites(N_SITES))
    d = cp.d
    mode = {"fd_sparse": E.GRAD_FD_SPARSE, "fd_dense": E.GRAD_FD_DENSE, "analytic": E.GRAD_ANALYTIC}[args.grad]
    cfg = E.hmc_config(grad_mode=mode, n_leapfrog=L)
    eng = E.Engine(cp, C, seed=1, chain_offset=rank * C, device=local_rank)
    stream = torch.cuda.current_stream()
    eng.set_stream(stream.cuda_stream)                    # kernels + torch events share one stream
    draws = torch.empty((K, d, C), dtype=torch.float64, device=f"cuda:{local_rank}")

    eng.hmc_init(cfg, Wn)
    draws.zero_()                                         # the output buffer's pages are touched before anything is timed
    # ---- clock spin-up first: throw-away transitions of the same kernel on a scratch engine (own state, own seed; the measured
    # engine is not touched), so that the W warmup and the K timed transitions run at the clocks the GPU settles to under this
    # load rather than on its way there (session set-up leaves it mostly idle).  The measured engine's W warmup transitions then
    # run directly before the timed region: its first launch (cold translation caches for its state and draw rows: +9 % on a
    # 20-transition launch) is a warmup launch whenever W > 0.
    if args.spinup > 0:
        scratch = E.Engine(cp, C, seed=987654321, chain_offset=rank * C, device=local_rank)
        scratch.set_stream(stream.cuda_stream)
        scratch.hmc_init(cfg, 0)
        t_sp = time.perf_counter()
        while time.perf_counter() - t_sp < args.spinup:
            scratch.hmc_step(4 * args.launch)
            torch.cuda.synchronize()
    # ---- untimed by the contract (reported separately): W adaptive warmup transitions
    warm_events = []
    t_warm = clock.region(lambda: warm_events.extend(stepped(torch, stream, lambda n, done: eng.hmc_step(n), Wn, args.launch)) if Wn > 0 else None)
    # ---- timed: exactly K sampling transitions (the scratch engine is freed afterwards: hipFree would put an idle gap between
    # the spin-up and the timed region)
    events = []
    dt = clock.region(lambda: events.extend(stepped(torch, stream, lambda n, done: eng.hmc_step(n, draws[done].data_ptr()), K, args.launch)))
    launch_ms, n_launch = full_launch_ms(events)
    if args.spinup > 0:
        scratch.close()

    # ---- after the timed region: the ONLY cross-chain step -- split R-hat / multichain ESS.  Each rank reduces its own draws
    # to per-chain moments on its GPU; the library all-gathers those and all-reduces the pooled lag sums over RCCL / xGMI
    # (fg_diag_rhat_ess, communicator created from an id that rank 0 obtains and torch.distributed's store hands out).
    t_diag = time.perf_counter()
    diag_path, comm = "library (single GPU)", None
    if world > 1 and (not one_device or os.environ.get("FG_BENCH_FORCE_NATIVE_RCCL") == "1"):   # (forced in the rehearsal mode: exercises the failure path, RCCL refuses two ranks on one device)
        ok = 1
        try:
            ids = [E.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(ids, src=0)
            # under a watchdog: this path has only ever run with one rank (one GPU per development box); a communicator that
            # cannot form must cost a bounded wait, not the run
            comm = call_with_timeout(lambda: eng.comm_init(world, rank, ids[0]), NATIVE_RCCL_TIMEOUT_S)
        except BaseException as ex:                              # every rank must take the same path
            sys.stderr.write(f"rank {rank}: RCCL communicator in the library failed ({ex!r}); falling back to torch.distributed\n")
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0 and comm is not None:
            E.comm_destroy(comm); comm = None
        diag_path = "library: ncclAllGather + ncclAllReduce" if comm is not None else "torch.distributed collectives + library combination"
    r = None
    if world == 1:
        r = eng.diag_rhat_ess(draws.data_ptr(), K, d, None)
    elif comm is not None:
        ok = 1
        try:
            r = call_with_timeout(lambda: eng.diag_rhat_ess(draws.data_ptr(), K, d, comm), NATIVE_RCCL_TIMEOUT_S)
        except BaseException as ex:
            sys.stderr.write(f"rank {rank}: fg_diag_rhat_ess over RCCL failed ({ex!r}); falling back to torch.distributed\n")
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            r, diag_path = None, "torch.distributed collectives + library combination (RCCL inside the library did not complete)"
        else:
            E.comm_destroy(comm)
    if r is not None:
        rhat, ess, n_chains_diag = r["r_hat"], (r["ess"] if K >= 4 else np.full(d, float("nan"))), r["chains"]
    else:
        if diag_path.startswith("library (single"):
            diag_path = "torch.distributed collectives + library combination"
        prov = D.EngineMoments(eng, draws.data_ptr(), K, d)
        cd = D.ChainDiagnostics(prov, device=None if one_device else coll_dev)
        rhat = cd.split_rhat()
        ess = cd.ess() if K >= 4 else np.full(d, float("nan"))
        n_chains_diag = cd.m
        prov.close()
    t_diag = time.perf_counter() - t_diag

    st = eng.hmc_stats()
    m = draws.mean(dim=(0, 2)).cpu().numpy()
    v = draws.var(dim=(0, 2)).cpu().numpy()
    _, tm, tv = W.normal_sites_truth(N_SITES)
    mean_err, var_err = float(np.abs(m - tm).max()), float(np.abs(v - tv).max())
    eng.close()
    del draws

    total_lf = world * C * K * L
    value = total_lf / dt
    # ---- roofline of the dominant kernel (k_hmc_sep_steps; k_hmc_stream_steps for the other gradient modes' fallbacks): algorithmic f64 work / HIP-event time
    n_stmt = 2 * N_SITES                                                      # S + O statements of the model
    evals_sparse = (2 * d * (L + 1)) * 2 + n_stmt                             # 2 signs x 2 dependent statements per coordinate per gradient + endpoint score
    evals_dense = (2 * d * (L + 1)) * n_stmt + n_stmt                         # 2 d (S + O) per gradient
    evals_dense_executed = (L + 1) * (n_stmt + 2 * 2 * d)                    # the dense kernel evaluates every statement once per gradient + the moved ones at +-h; the rest of the two scoring runs is additions
    evals = {E.GRAD_FD_SPARSE: evals_sparse, E.GRAD_FD_DENSE: evals_dense_executed, E.GRAD_ANALYTIC: d * (L + 1) * 2 + n_stmt}[mode]
    achieved_tflops = C * n_launch * evals * FLOPS_PER_NORMAL_LOGPDF / (launch_ms * 1e-3) / 1e12
    alg_bytes_per_launch = C * n_launch * (L * 32 * d + 8 * d + 16)           # 32 d B / leapfrog step (+ draw row, lj, eps)
    out = {
        "metric": "hmc_leapfrog_steps_per_sec", "value": value, "unit": "leapfrog-steps/s", "n_gpus": world,
        "steps": K, "warmup": Wn, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "C2-normal32: hmc_chain, 32-site conjugate Normal (x#i~N(0,1), y#i~N(x#i,0.5)=0.2i-1), "
                               f"{C} chains/GPU, L=16, HMCConfig::default", "chains_per_gpu": C, "n_sites": N_SITES,
                   "n_leapfrog": L, "grad": args.grad, "grad_note": "fd_sparse = the engine's default in the C ABI, Python and bench",
                   "transitions_per_launch": n_launch, "clock_spinup_seconds": args.spinup, "sharding": f"chains x{world}" if world > 1 else "single GPU"},
        "roofline": {"bound": "valu_f64", "achieved": achieved_tflops, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved_tflops / F64_VALU_PEAK_TFLOPS, "traffic": measured_traffic(C, n_launch, args.grad),
                     "kernel": "k_hmc_sep_steps", "avg_launch_ms": launch_ms,
                     "logpdf_evals_per_transition": evals, "flops_per_logpdf": FLOPS_PER_NORMAL_LOGPDF,
                     "note": "achieved = log-pdf evaluations the launch performs x 8 flops / HIP-event time; peak = f64 vector FMA peak "
                             "(2 flops/instr at 2.4 GHz) -- the arithmetic is unfused add/mul (reference rounding, 1 flop/instr) and the clock "
                             "sits near 2.1 GHz under f64 load, so ~36 TFLOP/s is the ceiling of this instruction mix.  traffic = measured "
                             "FETCH_SIZE + WRITE_SIZE per launch (separate rocprofv3 --pmc passes, profiles/)",
                     "executed": executed_from_pmc(C, n_launch, args.grad, launch_ms),
                     "dense_semantics": {"logpdf_evals_per_transition": evals_dense,
                                         "note": "SURVEY 8d's 2 d (S+O) log-pdfs per gradient are the two whole scoring runs of grad_log_joint; FG_GRAD_FD_DENSE adds "
                                                 "every one of their terms in order but evaluates only the densities that moved (see hmc_fd_dense), and the sparse "
                                                 "default never forms the terms that cancel in the reference's subtraction"},
                     "hbm_nominal": {"achieved": alg_bytes_per_launch / (launch_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                     "note": "SURVEY 8d algorithmic bytes (32 d B per leapfrog step, as if q and p round-tripped HBM) / time: NOT a "
                                             "claim -- q, p stay in registers for a whole trajectory and the kernel is not HBM bound"}},
        "incl_warmup": {"leapfrog_steps_per_sec": world * C * (K + Wn) * L / (dt + t_warm) if Wn > 0 else value,
                        "warmup_seconds": t_warm, "note": "SURVEY 8d counts leapfrog steps over warmup + sampling; `value` follows the bench contract (K timed "
                                                          "sampling transitions after W untimed warmup transitions)"},
        "check": {"posterior_mean_max_abs_err": mean_err, "posterior_var_max_abs_err": var_err,
                  "accept_rate": st.accept_rate, "mean_step_size": st.mean_step_size, "n_divergent": int(st.n_divergent),
                  "split_rhat_max": float(np.max(rhat)), "ess_min": float(np.min(ess)), "chains_in_rhat": int(n_chains_diag),
                  "diagnostics_seconds": t_diag, "diagnostics_path": diag_path,
                  "note": "statistics of the K timed draws themselves: with few steps / a short warmup they are NOT the 1e-3 evidence (a chain of "
                          "20 draws after 5 warmup transitions has not mixed) -- see `validity`"},
    }
    progress(rank, f"hmc leg done: {value:.4g} leapfrog-steps/s over {world} rank(s)")
    if not args.no_extras:
        out["mh"] = leg_mh(args, E, W, torch, clock, stream, world, rank, local_rank)
        progress(rank, "mh leg done")
        out["smc"] = leg_smc(args, E, W, torch, clock, stream, world, rank, local_rank)
        progress(rank, "smc leg done")
        if rank == 0:
            out["extras"] = extras(args, E, W, local_rank)
            progress(rank, "extras done")
            out["validity"] = validity(E, W, D, local_rank)
            progress(rank, "validity leg done")
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            gpu_dense = out.get("extras", {}).get("hmc_fd_dense_leapfrog_steps_per_sec")
            out["cpu_baseline"] = cpu_baseline_hmc(args, gpu_dense)
            if "mh" in out:
                out["mh"]["cpu_baseline"] = cpu_baseline_mh()
            if "smc" in out:
                out["smc"]["cpu_baseline"] = cpu_baseline_smc()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def leg_mh(args, E, W, torch, clock, stream, world, rank, dev):
    """The MCMC half of BASELINE's metric: adaptive_mcmc_chain on the reference's own bench model
    (benches/f_perf.rs:78-109: reference_model(20), 20 sample + 19 observe sites) at 65 536 chains per GPU; chain steps are
    counted over warmup + sampling: 200 adapting + 400 sampling steps, all timed."""
    C = args.chains
    cp = E.compile_model(W.reference_model(20))
    eng = E.Engine(cp, C, seed=1, chain_offset=rank * C, device=dev)
    eng.set_stream(stream.cuda_stream)
    nw, ns, per = 200, 400, 100
    eng.mh_init(nw)
    eng.mh_step(per)                                       # untimed: first-launch effects
    if args.spinup > 0:                                    # clock spin-up on a scratch engine, as in the HMC leg
        scratch = E.Engine(cp, C, seed=987654321, chain_offset=rank * C, device=dev)
        scratch.set_stream(stream.cuda_stream)
        scratch.mh_init(0)
        t_sp = time.perf_counter()
        while time.perf_counter() - t_sp < args.spinup:
            scratch.mh_step(4 * per)
            torch.cuda.synchronize()
    eng.mh_init(nw)
    events = []
    dt = clock.region(lambda: events.extend(stepped(torch, stream, lambda n, done: eng.mh_step(n), nw + ns, per)))
    launch_ms, n_launch = full_launch_ms(events)
    if args.spinup > 0:
        scratch.close()
    acc = eng.mh_stats().accept_rate
    eng.close()
    S, O = cp.S, cp.O
    bytes_per_step = 8 * S + 40                            # value row + 1 value + adaptation RMW + lw
    tflops = C * n_launch * (S + O) * FLOPS_PER_NORMAL_LOGPDF / (launch_ms * 1e-3) / 1e12
    return {"metric": "mh_chain_steps_per_sec", "value": world * C * (nw + ns) / dt, "unit": "chain-steps/s", "n_gpus": world,
            "accept_rate": acc,
            "config": {"workload": f"adaptive_mcmc_chain, reference_model(20) (benches/f_perf.rs:78-91: S=20, O=19), {C} chains/GPU, "
                                   f"{nw} adapting + {ns} sampling steps, all timed", "steps_per_launch": n_launch},
            "roofline": {"bound": "valu_f64", "achieved": tflops, "peak": F64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tflops / F64_VALU_PEAK_TFLOPS,
                         "traffic": measured_traffic_mh(C, nw, ns, n_launch) if C == CHAINS_PER_GPU else None, "kernel": "k_mh_mw_steps", "avg_launch_ms": launch_ms,
                         "note": "(S + O) log-pdfs x 8 flops per chain step / HIP-event time",
                         "hbm_nominal": {"achieved": C * n_launch * bytes_per_step / (launch_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "note": "SURVEY 8d: 8 S + 40 B per chain step; the value row lives in LDS across a launch"}},
            "published_reference": "65 k chain-steps/s/thread (15.3 us per transition, Apple Silicon; benches/f_perf.rs:24-28)"}


def leg_smc(args, E, W, torch, clock, stream, world, rank, dev):
    """C4: adaptive_smc, 1 048 576 particles, Systematic / ESS 0.5 / 3 rejuvenation moves (examples/smc_inference.rs:36-65).
    SMC does not shard without communication (next_beta, normalisation and resampling are global): N GPUs = N independent
    replicas with different seeds -- no collective is invented."""
    N = SMC_PARTICLES
    cp = E.compile_model(W.smc_normal())
    eng = E.Engine(cp, N, seed=42 + rank, device=dev)
    eng.set_stream(stream.cuda_stream)
    eng.smc_run(rejuvenation_steps=3, download=False)      # untimed: allocations, first-launch effects
    if args.spinup > 0:                                    # clock spin-up: the same run on a scratch population
        scratch = E.Engine(cp, N, seed=987654321 + rank, device=dev)
        scratch.set_stream(stream.cuda_stream)
        t_sp = time.perf_counter()
        while time.perf_counter() - t_sp < args.spinup:
            scratch.smc_run(rejuvenation_steps=3, download=False)
    res = {}
    dt = clock.region(lambda: res.update(eng.smc_run(rejuvenation_steps=3, download=False)))   # particles and weights stay in HBM
    if args.spinup > 0:
        scratch.close()
    eng.close()
    n_steps = len(res["betas"])
    moves = (res["n_model_runs"] - N) / 2
    S = cp.S
    per_particle_step = 24 + 1040 + 12 + 16 * S + 3 * (16 * S + 24)      # reweight + next_beta (65 passes x 16 B) + resample + gather + rejuvenation
    gbs = N * n_steps * per_particle_step / dt / 1e9
    return {"metric": "smc_particle_moves_per_sec", "value": world * moves / dt, "unit": "particle-moves/s", "n_gpus": world,
            "seconds_per_run": dt, "tempering_steps": n_steps, "log_evidence": res["log_evidence"], "log_evidence_closed_form": -1.9305103088617774,
            "config": {"workload": f"C4: adaptive_smc, {N} particles, Systematic / 0.5 / 3 rejuvenation moves, mu~N(0,1); y~N(mu,0.5)=1.5",
                       "sharding": "replicas only" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": measured_traffic_smc(),
                         "kernel": "whole fg_smc_run (next_beta bisection + reweight + scan + resample + gather + rejuvenation)",
                         "bytes_per_particle_per_tempering_step": per_particle_step,
                         "note": "SURVEY 8d algorithmic bytes x particles x tempering steps / wall time of the whole run (host-timed, launch gaps "
                                 "included); the 16 MB of (ll, lw) fit the L2 / Infinity Cache, so HBM is not what bounds the 65 ESS evaluations"}}


def extras(args, E, W, dev):
    """Side measurements on rank 0: the reference-verbatim dense finite difference, the opt-in analytic gradient and C5."""
    out = {}
    C = args.chains
    cp = E.compile_model(W.normal_sites(N_SITES))
    eng = E.Engine(cp, C, seed=1, device=dev)
    eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_FD_DENSE), 0)
    eng.hmc_step(10); eng.synchronize()
    t0 = time.perf_counter(); eng.hmc_step(50); eng.synchronize(); dt = time.perf_counter() - t0
    out["hmc_fd_dense_leapfrog_steps_per_sec"] = C * 50 * 16 / dt
    out["hmc_fd_dense_note"] = ("grad_log_joint verbatim (hmc.rs:304-329): every g_i is the difference of two whole log-joints.  The kernel performs the "
                                "additions of both full scoring runs (2 d (S + O) per gradient) but evaluates only the densities that moved: SURVEY 8d's "
                                "2 d (S + O) log-pdfs x 8 flops per gradient would be %.1f TFLOP/s at this rate and is NOT executed") % (
        C * 50 * ((2 * N_SITES * 17) * 2 * N_SITES + 2 * N_SITES) * FLOPS_PER_NORMAL_LOGPDF / dt / 1e12)
    eng.close()
    eng = E.Engine(cp, C, seed=1, device=dev)
    eng.hmc_init(E.hmc_config(grad_mode=E.GRAD_ANALYTIC), 0)
    eng.hmc_step(25); eng.synchronize()
    t0 = time.perf_counter(); eng.hmc_step(100); eng.synchronize(); dt = time.perf_counter() - t0
    out["hmc_analytic_leapfrog_steps_per_sec"] = C * 100 * 16 / dt
    eng.close()
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
    # C5 on one GPU: 4-component mixture (4 f64 + 64 usize sites, 64 observations), adaptive_mcmc_chain at 262 144 chains
    data, _ = W.mixture_data(64)
    cp5 = E.compile_model(W.mixture(data))
    eng = E.Engine(cp5, 262144, seed=1, device=dev)
    eng.mh_init(200)
    eng.mh_step(200); eng.synchronize()
    t0 = time.perf_counter(); eng.mh_step(200); eng.synchronize(); dt = time.perf_counter() - t0
    out["mh_mixture_262144_chain_steps_per_sec"] = 262144 * 200 / dt
    out["mh_mixture_hbm_nominal_gbs"] = 262144 * 200 * (8 * cp5.S + 40) / dt / 1e9
    out["mh_mixture_workload"] = f"C5: 4-component mixture, S={cp5.S} (4 f64 + 64 usize), O={cp5.O}, 262144 chains on one GPU"
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
    run_rank(args)
    if ABANDONED:                                                # a thread is stuck inside a collective: do not wait for it at interpreter exit
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
