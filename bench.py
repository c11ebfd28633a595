#!/usr/bin/env python3
"""bench.py -- slides/sec of the RRTEncoder forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C]

With --gpus N > 1 and no torch.distributed environment, bench.py re-executes itself under
``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`` (one rank per
GPU, RCCL); launched that way by hand it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
environment.  Rank 0 prints ONE JSON line.

--config selects the workload by its index in BASELINE.json ``configs`` (default 1, the config the metric is
quoted on; the N = 1 line of that config is what the driver records):
  0  encoder, one bag N=512 D=512 (the reference's CPU-runnable plumbing case), fp32
  1  encoder, N=9000 D=512 region_num=8, fp32                                   <- headline
  2  C16-R50 slide classifier: 1024 -> 512 fc + ReLU, encoder(epeg_k=15, crmsa_k=1, all_shortcut),
     DAttention, predictor; N=9000, bf16 autocast-class arithmetic (one C-ABI call per bag)
  3  survival long sequence: encoder N=30000 region_num=16, bf16, every GPU owns its bags
  4  TCGA-NSCLC mix: 64 bags, N ~ randint(3000, 15001) (seed 2021), epeg_k=21 crmsa_k=5; the batch is split over
     the ranks by cost (sharding.assign_bags, longest-processing-time first) and each rank runs its share through
     the batch-of-bags executor; a step = one pass over the whole batch ("scaling": "strong")
In configs 0-3 a step = one batch of `--streams` (default 4; 3 for bf16 bags of > 12 k tokens) x `--bags-per-stream` (default 4) independent
bags per GPU: `--streams` bags in flight, each an ordinary forward on its own HIP stream with its own workspace (bags are
independent units), and every rank owns its own bags ("scaling": "weak").  No data-path collective anywhere: RCCL carries the barrier and a MAX of the elapsed time.

Besides the contract fields the JSON line carries
  roofline     -- the dominant kernel (the fused R-MSA kernel): algorithmic FLOPs per launch / its average
                  duration, measured live with HIP events that librrt_hip records on the launch stream inside
                  the timed region (config 2: in an untimed pass of the same encoder; the one-call classifier
                  entry takes no event array; config 4: summed over the rank's bags in an untimed pass);
                  `traffic` = HBM-side bytes per launch of that kernel from the rocprofv3 PMC passes of THIS config,
                  read from the newest profiles/rNN_traffic.json (tools/pmc_to_traffic.py; never a typed-in constant);
                  `frac` follows from the HIP-event UNION of the launches' intervals (several bags in flight: launches of
                  different streams overlap) when that is consistent with the line's own ms_per_step, and is never above
                  what ms_per_step allows; `frac_lower_bound` = FLOPs of the step's launches / ms_per_step / peak;
  roofline_kernels -- every stage of one bag's forward (LN + partition, fused R-MSA, out-projection, CR-MSA logits +
                  combine, the representatives' MSA, dispatch + LayerNorm) with ITS bound: MFMA TFLOP/s for the matrix
                  kernels, HBM GB/s over the stage's algorithmic bytes for the streaming ones; stage boundaries from the
                  library's event marks in an untimed one-bag-in-flight pass;
  cpu_baseline -- the oracle's eager torch-CPU port of the reference op sequence
                  (oracle/rrt_oracle.py::forward_eager) timed on this box's host cores over a bounded sample
                  of the same workload (rank 0, N=1 only).
--dtype overrides a config's arithmetic (f32 | bf16 | f16 | f32x3 = fp32 in / out with the big products emulated on the bf16
matrix cores, RRT_COMPUTE_F32X3); the default run also reports bf16 and f32x3 as extra records (`amp_bf16`, `f32x3`).
--stub-cpu replaces the GPU workload by a tiny CPU one over gloo: the rank logic (spawn, sharding, barrier,
MAX over ranks, the JSON line) then runs without a GPU -- tests/test_multiproc_cpu.py drives it.
"""
import argparse
import contextlib
import ctypes as C
import json
import os
import socket
import sys
import time

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order.  With the
# default, the streams RCCL creates under torch.distributed.run shift that mapping so that the bags' two
# streams land on ONE hardware queue and serialise (measured: 3.72 k slides/s instead of 4.33 k, exactly the
# one-stream rate).  16 queues keep every stream of this process on its own queue in both launch modes (8 still lost 4 %
# under torch.distributed.run in round 2: 4.44 k vs 4.57 k with 16, 4.63 k without RCCL in the process).
# Must be set before the HIP runtime initialises, i.e. before `import torch`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DIM = 512
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}   # MI355X_MICROARCH.md: dense MFMA peaks (f32 in / bf16 in)
PEAK_HBM_GBS = 8000.0                          # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec; ~6.3 TB/s achievable)
FRESH_BYTES = 512 << 20                        # --fresh-inputs: distinct input bytes in rotation (2 x the 256 MiB Infinity Cache)
def _newest(stem):
    """profiles/rNN_<stem> of the latest round that has one (the files are written by tools/, never typed in)"""
    for rnd in ("r06", "r05", "r04", "r03"):
        f = os.path.join(ROOT, "profiles", f"{rnd}_{stem}")
        if os.path.exists(f):
            return f
    return os.path.join(ROOT, "profiles", f"r06_{stem}")


TRAFFIC_FILE = _newest("traffic.json")             # tools/pmc_to_traffic.py, from the PMC passes
ROCPROF_FILE = _newest("rocprof_dominant.json")    # tools/rocprof_union.py, from --kernel-trace runs


def rocprof_record(config, dtype, streams):
    """The dominant kernel's duration in the committed rocprofv3 kernel trace of this (config, dtype, streams): mean
    dispatch duration and union-of-intervals per launch (tools/rocprof_union.py writes the file), or None."""
    try:
        with open(ROCPROF_FILE) as fh:
            r = json.load(fh).get(f"c{config}_{dtype}_s{streams}")
    except (OSError, ValueError):
        return None
    if not r:
        return None
    r = dict(r)
    r["source"] = os.path.relpath(ROCPROF_FILE, ROOT)
    return r



# kernels that ARE the R-MSA core of a layer (one launch per layer and forward); "rmsa_fused16_kernel<4" is not in the
# list: at the bench's bag sizes that instantiation only runs CR-MSA's inner MSA (k "regions" of 64 representatives)
RMSA_CORE_KERNELS = ("rmsa_pair16", "rmsa_fused16_kernel<1", "rmsa_fused16_kernel<6", "rmsa_fused16_kernel<7",
                     "rmsa_fused16_kernel<8", "rmsa_fused16_kernel<9", "rmsa_fused_kernel", "rmsa_fused_x3")


def traffic_table(config, dtype):
    """{kernel: {hbm_bytes, dispatches, ...}} of this (config, dtype) from the tracked PMC summary, or None."""
    try:
        with open(TRAFFIC_FILE) as fh:
            return json.load(fh).get(f"c{config}_{dtype}", {}).get("kernels")
    except (OSError, ValueError):
        return None


def traffic_of(table, patterns, per_forward_of=None):
    """HBM-side bytes of the kernels whose names contain one of `patterns`: per launch (one kernel), or -- with
    per_forward_of = the name pattern of a kernel that runs once per forward -- summed per forward."""
    if not table:
        return None
    hits = {k: v for k, v in table.items() if any(p in k for p in patterns)}
    if not hits:
        return None
    if per_forward_of is None:
        k = max(hits, key=lambda n: hits[n]["dispatches"])
        return {"bytes": hits[k]["hbm_bytes"], "kernel": k, "source": os.path.relpath(TRAFFIC_FILE, ROOT)}
    base = [v["dispatches"] for k, v in table.items() if any(p in k for p in per_forward_of)]
    if not base or max(base) == 0:
        return None
    nfwd = float(max(base))
    return {"bytes": int(sum(v["hbm_bytes"] * v["dispatches"] / nfwd for v in hits.values())),
            "kernels": sorted(hits), "source": os.path.relpath(TRAFFIC_FILE, ROOT)}

CONFIGS = {
    0: dict(kind="encoder", n=512, dtype="f32", enc=dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8),
            label="BASELINE configs[0]: RRTEncoder(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8).eval() forward, "
                  "one step = ONE rrt_mil_amd.RRTEncoder.forward_bags call of `bags_per_step` device-resident bags N=512 D=512, "
                  "`streams_per_gpu` of them in flight"),
    1: dict(kind="encoder", n=9000, dtype="f32", enc=dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8),
            label="BASELINE configs[1]: RRTEncoder(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8).eval() forward, "
                  "one step = ONE call of the drop-in module's batch entry, rrt_mil_amd.RRTEncoder.forward_bags(bags, streams=S, "
                  "outs=...), over `bags_per_step` device-resident bags N=9000 D=512 (`streams_per_gpu` bags in flight, each an "
                  "ordinary forward with its own workspace; --raw-loop: the C-ABI entry point per bag instead); fp32, closed-form "
                  "weights"),
    2: dict(kind="mil", n=9000, dtype="bf16", input_dim=1024, streams=4,
            enc=dict(mlp_dim=512, epeg_k=15, crmsa_k=1, region_num=8, all_shortcut=True),
            label="BASELINE configs[2]: C16-R50 RRTMIL(input_dim=1024, epeg_k=15, crmsa_k=1, all_shortcut=True).eval() "
                  "forward, N=9000 x 1024 non-negative features -> logits (rrt_mil_forward_f32: fc + ReLU, encoder, "
                  "DAttention pooling, predictor), bf16 autocast-class arithmetic"),
    3: dict(kind="encoder", n=30000, dtype="bf16", enc=dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=16),
            label="BASELINE configs[3]: survival long sequence, RRTEncoder(region_num=16, epeg_k=15, crmsa_k=3).eval() "
                  "forward_bags call per step over device-resident bags N=30000 D=512 owned by each GPU, bf16 autocast-class arithmetic"),
    4: dict(kind="mix", n=None, dtype="bf16", n_bags=64, streams=4, enc=dict(mlp_dim=512, epeg_k=21, crmsa_k=5, region_num=8),
            label="BASELINE configs[4]: TCGA-NSCLC-R50 encoder (epeg_k=21, crmsa_k=5), one batch of 64 device-resident "
                  "bags with N ~ randint(3000, 15001) (seed 2021), split over the ranks by cost (LPT), each rank's share "
                  "through the batch-of-bags executor"),
}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` as the driver invokes it: become N ranks on this node."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def mix_sizes(n_bags):
    import numpy as np
    return [int(v) for v in np.random.RandomState(2021).randint(3000, 15001, size=n_bags)]


def flops_total(n, enc_cfg):
    """Algorithmic FLOPs per bag, SURVEY.md §8(d)."""
    from rrt_mil_amd.geometry import region_grid
    g, g8 = region_grid(n, enc_cfg["region_num"]), region_grid(n, 8)
    k, h, ek, D = enc_cfg["crmsa_k"], 8, enc_cfg["epeg_k"], DIM
    return (8 * g.Np * D * D + 4 * g.Np * g.P * D + 2 * g.Np * g.P * h * ek
            + 6 * g8.Np * D * k + 8 * k * 64 * D * D + 4 * k * 64 * 64 * D)


N_PARAMS = 2105984          # RRTEncoder(mlp_dim=512, ...): SURVEY.md section 8(a1); crmsa_k only moves phi (512 x k)


def bytes_total(n, in_dim=0):
    """Algorithmic bytes per bag, SURVEY.md section 8(d): B = (2 N D + n_params) x 4 -- the bag read once, the output written
    once, the weights once.  The boundary tensors are fp32 in every arithmetic mode (the reference's autocast keeps the
    module's input and output in fp32), so the 16-bit modes are priced against the same B.  Classifier (config 2): the
    N x in_dim features in, logits out, the fc weight once."""
    if in_dim:
        return (n * in_dim + in_dim * DIM + DIM + N_PARAMS) * 4
    return (2 * n * DIM + N_PARAMS) * 4


def traffic_per_forward(config, dtype, kind):
    """HBM-side bytes of ONE forward summed over its kernels from the tracked PMC table (None without one).  A kernel's
    launches per forward = its dispatch count over that of the dispatch kernel (once per forward); kernels that ran for a
    quarter to three quarters of the forwards are the classifier's own stages (the table's run also holds encoder-only
    passes) and count once; rarer ones (weight casts, torch's finiteness checks) are not part of a steady-state forward."""
    table = traffic_table(config, dtype)
    if not table:
        return None
    base = max([v["dispatches"] for k, v in table.items() if "crmsa_dispatch_ln" in k] or [0])
    if base == 0:
        return None
    total, used = 0.0, []
    for k, v in table.items():
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        r = v["dispatches"] / base
        lpf = round(r) if r >= 0.75 else (1 if (kind == "mil" and r >= 0.25) else 0)
        if kind == "mix":
            lpf = r                      # mixed sizes: every kernel's share of the 64-bag batch, averaged per bag
        if lpf:
            total += v["hbm_bytes"] * lpf
            used.append(k.split("(")[0][:48])
    return {"bytes": int(total), "kernels": len(used), "source": os.path.relpath(TRAFFIC_FILE, ROOT)}


def whole_path_roofline(slides_per_s_per_gpu, flops_per_bag, bytes_per_bag, dtype, traffic):
    """The whole forward against BOTH roofs (T_roof = max(F / peak_matrix(dtype), B / peak_HBM)):
    what the line's own rate makes of the algorithmic FLOPs and bytes, and what the PMC-counted traffic says."""
    mpeak = PEAK_TFLOPS["bf16" if dtype in ("bf16", "f16", "f32x3") else "f32"]
    t = 1.0 / slides_per_s_per_gpu
    t_m, t_h = flops_per_bag / (mpeak * 1e12), bytes_per_bag / (PEAK_HBM_GBS * 1e9)
    rec = {"flops_per_bag": flops_per_bag, "algorithmic_bytes_per_bag": bytes_per_bag,
           "mfma": {"achieved": round(flops_per_bag / t / 1e12, 2), "peak": mpeak, "unit": "TFLOP/s",
                    "frac": round(t_m / t, 4)},
           "hbm": {"achieved": round(bytes_per_bag / t / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                   "frac": round(t_h / t, 4)},
           "bound": "mfma" if t_m >= t_h else "hbm", "frac": round(max(t_m, t_h) / t, 4)}
    if traffic:
        tb = traffic["bytes"]
        rec["traffic"] = {"bytes_per_bag": tb, "over_algorithmic": round(tb / bytes_per_bag, 2),
                          "rate_GBs": round(tb / t / 1e9, 1), "frac_of_hbm_peak": round(tb / t / 1e9 / PEAK_HBM_GBS, 4),
                          "counter": "TCC_EA requests (2 x FETCH_SIZE + WRITE_SIZE, one bag in flight): includes what the "
                                     "256 MB Infinity Cache (MALL) served, i.e. an upper bound on DRAM bytes",
                          "source": traffic["source"]}
    return rec


def fused_flops(n, enc_cfg, with_proj=False):
    """The fused R-MSA kernel's share: qkv projection [Np, D] x [3D, D]^T + Q K^T + A V per (region, head) -- and, where
    the out-projection runs as a later phase of the same launch (rrt_encoder_plan: RRT_PLAN_FUSED_PROJ), [Np, D] x [D, D]^T."""
    from rrt_mil_amd.geometry import region_grid
    g = region_grid(n, enc_cfg["region_num"])
    return 2.0 * g.Np * (3 * DIM) * DIM + 4.0 * g.Np * g.P * DIM + (2.0 * g.Np * DIM * DIM if with_proj else 0.0), g


def plan_flags(enc, n):
    """rrt_encoder_plan of an RRTEncoder's current descriptor for a bag of n tokens"""
    from rrt_mil_amd import _lib
    fl = C.c_int32(0)
    _lib.check(_lib.load().rrt_encoder_plan(C.byref(enc._desc), n, C.byref(fl)), "rrt_encoder_plan")
    return fl.value


class HipEvents:
    """Raw hipEvent_t's (libamdhip64 via ctypes) -- torch.cuda.Event cannot be recorded
    from inside the C ABI call."""

    def __init__(self):
        self.hip = C.CDLL("libamdhip64.so")
        self.hip.hipEventCreate.argtypes = [C.POINTER(C.c_void_p)]
        self.hip.hipEventElapsedTime.argtypes = [C.POINTER(C.c_float), C.c_void_p, C.c_void_p]
        self.hip.hipEventDestroy.argtypes = [C.c_void_p]

    def create(self):
        ev = C.c_void_p()
        rc = self.hip.hipEventCreate(C.byref(ev))
        assert rc == 0, f"hipEventCreate -> {rc}"
        return ev

    def elapsed_ms(self, a, b):
        ms = C.c_float()
        rc = self.hip.hipEventElapsedTime(C.byref(ms), a, b)
        assert rc == 0, f"hipEventElapsedTime -> {rc}"
        return ms.value


def cpu_baseline(n_tokens, enc_cfg, budget_s=22.0):
    """Reference-equivalent CPU path (oracle port) on this host: bounded sample.  The eager op
    sequence scales poorly past a few dozen threads (many small aten ops), so a short probe picks
    the thread count at which the reference path is FASTEST before the timed sample."""
    import numpy as np
    import torch
    from oracle import rrt_oracle  # the only place bench.py touches oracle/
    from rrt_mil_amd import synth
    state = synth.encoder_state(**{k: v for k, v in enc_cfg.items() if k in ("mlp_dim", "epeg_k", "crmsa_k")})
    st = {k: torch.from_numpy(v) for k, v in state.items()}
    x = torch.from_numpy(synth.bag(n_tokens, DIM))
    ncpu = os.cpu_count() or torch.get_num_threads()
    cands = sorted({c for c in (ncpu, ncpu // 2, ncpu // 4, 32, 16, 8) if 1 <= c <= ncpu}, reverse=True)
    probe = {}
    for c in cands:
        torch.set_num_threads(c)
        rrt_oracle.forward_eager(x, st, enc_cfg)         # warm-up at this thread count
        t0 = time.perf_counter()
        rrt_oracle.forward_eager(x, st, enc_cfg)
        probe[c] = time.perf_counter() - t0
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    times = []
    t_end = time.perf_counter() + budget_s * 0.6
    while len(times) < 5 or (time.perf_counter() < t_end and len(times) < 200):
        t0 = time.perf_counter()
        rrt_oracle.forward_eager(x, st, enc_cfg)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    cpu_model = None
    try:
        with open("/proc/cpuinfo") as fh:
            cpu_model = next((ln.split(":", 1)[1].strip() for ln in fh if ln.startswith("model name")), None)
    except OSError:
        pass
    rec = {"value": round(1.0 / med, 3), "unit": "slides/s", "cores": cores, "kind": "port",
           # the box this baseline was taken on (it differs from lease to lease: 6.0-9.6 slides/s over round 5's boxes)
           "box": {"cpu_model": cpu_model, "hardware_threads": ncpu,
                   "thread_probe_ms": {str(c): round(probe[c] * 1e3, 1) for c in cands}},
           "sample": f"{len(times)} bags of N={n_tokens} D={DIM} (median {med * 1e3:.1f} ms/bag), "
                     f"oracle/rrt_oracle.py::forward_eager (same aten op sequence as the reference, "
                     f"bit-identical to it in the build container), torch {torch.__version__} CPU, "
                     f"{cores} of {ncpu} hardware threads (fastest of "
                     + ", ".join(f"{c}: {probe[c] * 1e3:.0f} ms" for c in cands) + ")"}
    # the port's time relative to the REAL reference, measured where both exist (the build
    # container, tools/port_vs_reference.py, interleaved runs on the same cores)
    try:
        with open(os.path.join(ROOT, "profiles", "port_vs_reference_container.json")) as fh:
            cal = json.load(fh)
        rec["port_vs_oracle"] = {"ratio": cal["ratio_port_over_reference"], "port_ms": cal["port_ms"],
                                 "reference_ms": cal["reference_ms"], "threads": cal["threads"],
                                 "where": "build container (the reference never travels to the GPU box); "
                                          "tools/port_vs_reference.py"}
    except (OSError, KeyError, ValueError):
        rec["port_vs_oracle"] = None
    return rec


# ------------------------------------------------------------------------------------ workloads
class StubWorkload:
    """--stub-cpu: the rank logic without a GPU.  Config 4 semantics: a batch of mixed-size bags split by
    sharding.assign_bags; 'processing' a bag is a small CPU matmul proportional to its size."""

    def __init__(self, args, rank, world, dev):
        import torch
        from rrt_mil_amd import sharding
        self.torch = torch
        cfg = CONFIGS[args.config]
        self.sizes = mix_sizes(cfg.get("n_bags", 8)) if cfg["kind"] == "mix" else [cfg["n"]] * (world * 2)
        self.mine = sharding.assign_bags(self.sizes, world, **self._cost_kw(cfg))[rank]
        self.units_global = len(self.sizes)
        self.scaling = "strong" if cfg["kind"] == "mix" else "weak"
        self.w = torch.ones(64, 64)
        self.acc = 0.0
        self.extra = {"bags_this_rank": len(self.mine)}

    @staticmethod
    def _cost_kw(cfg):
        e = cfg["enc"]
        return dict(region_num=e["region_num"], epeg_k=e["epeg_k"], crmsa_k=e["crmsa_k"])

    def step(self, i, timed):
        for b in self.mine:
            x = self.torch.full((max(1, self.sizes[b] // 1000), 64), 1.0 / 64)
            self.acc += float((x @ self.w).sum())

    def sync(self):
        pass

    def finish(self, args, world, rank, elapsed):
        return {}


class EncoderWorkload:
    """configs 0, 1, 3 (every rank owns S bags in flight) and 2 (the classifier around the encoder)."""

    def __init__(self, args, rank, world, dev):
        import torch
        from rrt_mil_amd import RRTEncoder, RRTMIL, _lib, synth
        self.torch, self._lib = torch, _lib
        cfg = self.cfg = CONFIGS[args.config]
        if os.environ.get("RRT_BENCH_N") and cfg["kind"] == "encoder":      # (experiments only: another bag size for the same line)
            cfg = self.cfg = dict(cfg, n=int(os.environ["RRT_BENCH_N"]))
        self.dev, self.n, self.enc_cfg = dev, cfg["n"], cfg["enc"]
        self.dtype = args.dtype or cfg["dtype"]
        self.compute = {"f32": _lib.COMPUTE_F32, "bf16": _lib.COMPUTE_BF16, "f16": _lib.COMPUTE_F16,
                        "f32x3": _lib.COMPUTE_F32X3}[self.dtype]
        self.S = S = max(1, args.streams)
        # a step = one batch of synthetic bags = S bags in flight x R bags per stream (R back to back on every stream): the
        # timed region is bracketed by device syncs, so its first bags start in lockstep and its last ones drain alone --
        # with R = 1 a 20-step region (80 bags) read 2 % under a 200-step one; with R = 4 the same 20 steps are 320 bags
        # Round 6: the timed region of the encoder configs goes through the drop-in itself -- one
        # `rrt_mil_amd.RRTEncoder.forward_bags(bags, streams=S, outs=...)` call per step, from a caller under its own
        # `with torch.cuda.stream(s):` (the asynchronous use INTEGRATION.md section 4 documents) -- so a step is one CALL of
        # S x R bags, R = 16 by default: the call's fork / join is a barrier across the bag streams, paid once per step.
        # `--raw-loop` (and the classifier of config 2, one C-ABI call per slide) keeps the rounds 1-5 region: the C-ABI entry
        # point called per bag on S streams, R = 4.
        self.via = "raw_loop" if (getattr(args, "raw_loop", False) or cfg["kind"] == "mil") else "forward_bags"
        self.Rr = 4                                   # bags per stream and step of the raw C-ABI loop
        # (64 bags per call read 2 % under the raw loop -- 5198 against 5304 slides/s fp32, same box -- and 256 bags 0.5 %:
        #  profiles/r06_probe1.txt, r06_p2_lines.txt; long bags keep 16 per stream: a call of 48 N=30000 bags is already 8 ms)
        self.R = R = max(1, getattr(args, "bags_per_stream", 0) or (self.Rr if self.via == "raw_loop" else 64 if cfg["n"] <= 12000 else 16))
        if self.via == "raw_loop":
            self.Rr = R
        self.raw_steps = args.steps if self.via == "raw_loop" else 20      # steps of the raw pass (forward_bags runs: untimed, after the region)
        self.units_global = world * S * R
        self.scaling = "weak"
        self.extra, self.fresh = {}, False
        self.lib = _lib.load()
        self.hev = HipEvents()
        tstreams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(S - 1)]
        if os.environ.get("RRT_BENCH_NO_NULL") == "1":       # (experiment: no bag on the process's default stream)
            tstreams = [torch.cuda.Stream(dev) for _ in range(S)]
        self._tstreams = tstreams
        self.streams = [t.cuda_stream for t in tstreams]
        self.mil = None
        if cfg["kind"] == "mil":
            mcfg = dict(input_dim=cfg["input_dim"], n_classes=2, **{k: v for k, v in self.enc_cfg.items()
                                                                     if k not in ("mlp_dim", "region_num")})
            mil = RRTMIL(**mcfg).eval()
            mst = synth.mil_state(input_dim=cfg["input_dim"], n_classes=2, epeg_k=self.enc_cfg["epeg_k"],
                                  crmsa_k=self.enc_cfg["crmsa_k"])
            mil.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in mst.items()}, strict=True)
            self.mil = mil.to(dev)
            self.enc = self.mil.online_encoder
            self.enc.compute_dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16,
                                      "f32x3": "f32x3"}[self.dtype]
            self.bags = [torch.from_numpy(synth.bag(self.n, cfg["input_dim"], tag=f"bench/mil/r{rank}/b{i}",
                                                    nonneg=True)).to(dev) for i in range(4)]
            self.mdesc, self.mw = self.mil._mil_desc(cfg["input_dim"], solo=(S == 1)), self.mil._mil_weights()
            need = C.c_size_t()
            _lib.check(self.lib.rrt_mil_workspace_size(C.byref(self.mdesc), self.n, C.byref(need)), "mil workspace")
            self.wss = [torch.empty(need.value, dtype=torch.uint8, device=dev) for _ in range(S)]
            self.outs = [torch.empty(2, dtype=torch.float32, device=dev) for _ in range(S)]
            self._fresh(args, cfg["input_dim"], nonneg=True)
        else:
            state = synth.encoder_state(**{k: v for k, v in self.enc_cfg.items() if k != "region_num"})
            enc = RRTEncoder(**self.enc_cfg).eval()
            enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()}, strict=True)
            self.enc = enc.to(dev)
            nb = 4 if self.n <= 12000 else 2
            self.bags = [torch.from_numpy(synth.bag(self.n, DIM, tag=f"bench/r{rank}/b{i}")).to(dev) for i in range(nb)]
            need = self.enc._workspace(self.n, dev).numel()
            self.wss = [torch.empty(need, dtype=torch.uint8, device=dev) for _ in range(S)]   # one workspace per bag in flight
            self.outs = [torch.empty_like(self.bags[0]) for _ in range(S)]
            self._fresh(args, DIM, nonneg=False)
        self.enc._desc.compute = self.compute
        self.enc._desc.solo = int(S == 1)        # scheduling hint: with S > 1 the bags share the GPU (rrt_encoder_desc.solo)
        if os.environ.get("RRT_BENCH_SOLO") in ("0", "1"):     # (experiments only)
            self.enc._desc.solo = int(os.environ["RRT_BENCH_SOLO"])
        self.w = self.enc._weights()
        # one event pair per launch of the dominant kernel over a WINDOW of consecutive steps in the middle of the timed
        # region, every stream (an event pair costs the stream two marker packets: measured ~5 us per forward, 2 % of an
        # fp32 bag and 10 % of a bf16 one -- so the window is 12 bags per stream, not the whole region; one step of 16 launches
        # read anywhere between 0.27 and 0.35 ms run to run).  Consecutive steps, all
        # streams: the launches' intervals can then be merged into the time during which the kernel was running at all.
        self.ev_win = min(max(2, (12 if self.raw_steps >= 100 else 4) // self.Rr), self.raw_steps)   # (a 20-step run: two steps)
        self.ev_w0 = (self.raw_steps - self.ev_win) // 2
        self.ev_pairs = [(self.hev.create(), self.hev.create()) for _ in range(self.ev_win * S * self.Rr)]
        self.ev_arr = (C.c_void_p * _lib.EV_COUNT)()
        # optional phase gate (RRT_BENCH_GATE=1): the bags' MFMA-bound R-MSA cores take turns instead of time-slicing.
        # Off by default since round 2: with the denser kernels free-running streams are faster at every S
        # (fp32 S=2: 4.53 k vs 4.40 k slides/s, bf16: 12.5 k vs 11.7 k -- DESIGN.md section 5)
        self.gate = C.c_void_p()
        want_gate = os.environ.get("RRT_BENCH_GATE") == "1"
        if want_gate and self.mil is None:
            _lib.check(self.lib.rrt_phase_gate_create(C.byref(self.gate)), "phase gate")
        self._w16_mode = [None] * S      # compute mode of the 16-bit weight images in each stream's workspace
        if self.via == "forward_bags":
            self.enc.compute_dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16, "f32x3": "f32x3"}[self.dtype]
            self.caller = torch.cuda.Stream(dev)      # the caller's stream of every forward_bags call
            nb = S * R
            self.call_bags = [self.bags[j % len(self.bags)] for j in range(nb)]
            # one output per bag of the call (what a user gets); --fresh-inputs: also one distinct input per bag (self._fresh)
            self.call_outs = ([self.outs[j] for j in range(nb)] if self.fresh else [torch.empty_like(self.bags[0]) for _ in range(nb)])
            for o in self.call_outs:
                o.zero_()                # first touch outside the timed region
            self.outs_raw = self.call_outs[:S] if not self.fresh else self.outs
        else:
            self.outs_raw = self.outs

    def _fresh(self, args, width, nonneg):
        """--fresh-inputs: >= FRESH_BYTES of distinct input bags (the four synthetic ones + device-generated normal bags of the
        same shape) and, for the encoder configs, one output buffer per bag -- the rotation then touches more than twice the
        Infinity Cache between two uses of a buffer, so neither a bag nor its output is cache-resident when its forward starts."""
        self.fresh = bool(getattr(args, "fresh_inputs", False))
        if not self.fresh:
            return
        torch = self.torch
        bag_bytes = self.n * width * 4
        want = max(len(self.bags), 2 * self.S * self.Rr, self.S * self.R if self.via == "forward_bags" else 0,
                   -(-FRESH_BYTES // bag_bytes))
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(2021)
        while len(self.bags) < want:
            b = torch.randn(self.n, width, device=self.dev, generator=gen)
            self.bags.append(b.relu_() if nonneg else b)
        if self.mil is None:
            self.outs = [torch.empty_like(b) for b in self.bags]
            for o in self.outs:
                o.zero_()                  # first touch (page mapping) outside the timed region
        self.extra["fresh_inputs"] = {"distinct_bags": len(self.bags), "input_mb": round(len(self.bags) * bag_bytes / 1e6, 1),
                                      "distinct_outputs": len(self.outs) if self.mil is None else None}

    def _mark(self, a, b):
        for j in range(self._lib.EV_COUNT):
            self.ev_arr[j] = None
        self.ev_arr[self._lib.EV_LN_PARTITION], self.ev_arr[self._lib.EV_QKV] = a, b
        return self.ev_arr

    def step(self, i, timed):
        if self.via == "forward_bags":
            with self.torch.cuda.stream(self.caller):
                self.enc.forward_bags(self.call_bags, streams=self.S, outs=self.call_outs)
            return
        self.step_raw(i, timed)

    def step_raw(self, i, timed):
        """S x Rr bags through the C-ABI entry point, one call per bag, S streams (the timed region of rounds 1-5)."""
        lib, _lib = self.lib, self._lib
        for r_, s_ in ((r_, s_) for r_ in range(self.Rr) for s_ in range(self.S)):
            bi = ((i * self.Rr + r_) * self.S + s_) % len(self.bags)
            x = self.bags[bi]
            y = self.outs_raw[bi] if (self.fresh and self.mil is None) else self.outs_raw[s_]
            if self.mil is not None:
                mode = self.mdesc.enc.compute     # (as RRTMIL.forward_bag: the 16-bit weight images stay in the workspace)
                self.mdesc.enc.weights16_valid = int(mode != _lib.COMPUTE_F32 and self._w16_mode[s_] == mode)
                self._w16_mode[s_] = mode
                rc = lib.rrt_mil_forward_f32(C.byref(self.mdesc), C.byref(self.mw), x.data_ptr(), self.outs[s_].data_ptr(),
                                             None, 0, None, self.n, self.wss[s_].data_ptr(), self.wss[s_].numel(),
                                             self.streams[s_])
                _lib.check(rc, "rrt_mil_forward_f32")
                continue
            # mark the dominant kernel: [after LN+partition, after the fused R-MSA core]
            evs = (self._mark(*self.ev_pairs[((i - self.ev_w0) * self.Rr + r_) * self.S + s_])
                   if (timed and self.ev_w0 <= i < self.ev_w0 + self.ev_win) else None)
            # reduced-precision modes: this stream's workspace keeps the 16-bit weight images of the (unchanged)
            # weights from its first call on, as rrt_mil_amd.RRTEncoder does between forwards (weights16_valid)
            mode = self.enc._desc.compute
            self.enc._desc.weights16_valid = int(mode != _lib.COMPUTE_F32 and self._w16_mode[s_] == mode)
            self._w16_mode[s_] = mode
            rc = lib.rrt_encoder_forward_gated_f32(C.byref(self.enc._desc), C.byref(self.w), x.data_ptr(),
                                                   y.data_ptr(), self.n, self.wss[s_].data_ptr(),
                                                   self.wss[s_].numel(), self.streams[s_], self.gate, evs)
            _lib.check(rc, "forward")
        self.enc._desc.weights16_valid = 0

    def sync(self):
        self.torch.cuda.synchronize()

    def isolated_fused_ms(self, reps=10):
        """the dominant kernel alone on the chip (one bag in flight), untimed pass"""
        import numpy as np
        torch = self.torch
        enc = self.enc
        x = self.bags[0]
        if self.mil is not None:      # the encoder's own input: any [N, 512] buffer
            x = torch.from_numpy(__import__("rrt_mil_amd").synth.bag(self.n, DIM, tag="bench/iso")).to(self.dev)
        if self.mil is not None:
            need = enc._workspace(self.n, self.dev).numel()
            ws, y = torch.empty(need, dtype=torch.uint8, device=self.dev), torch.empty_like(x)
        else:                         # stream 0's own buffers (idle now)
            ws, y = self.wss[0], self.outs[0]
        # forwards enqueued back to back on one stream (as in a one-bag-in-flight loop): with a host sync between them
        # the event pair also times ~15 us of dispatch latency of an empty queue
        # ... and an event pair on EVERY forward costs the stream ~5 us of marker packets per forward, part of it inside
        # the interval: every fourth forward carries one, as in the timed region (agrees with rocprofv3's duration)
        pairs = [(self.hev.create(), self.hev.create()) for _ in range(reps)]
        lead = 150                                    # the host runs well ahead of the GPU, and the clocks are back up
        for i in range(lead + 4 * reps):                # after the idle gap that follows the timed region
            evs = self._mark(*pairs[(i - lead) // 4]) if i >= lead and (i - lead) % 4 == 0 else None
            self._lib.check(self.lib.rrt_encoder_forward_events_f32(C.byref(enc._desc), C.byref(self.w), x.data_ptr(),
                                                                    y.data_ptr(), self.n, ws.data_ptr(), ws.numel(),
                                                                    self.streams[0], evs), "forward")
        torch.cuda.synchronize()
        return float(np.median([self.hev.elapsed_ms(a, b) for a, b in pairs]))

    def staged_pass(self, reps=12):
        """One bag in flight, every stage boundary marked (librrt_hip records the events on the launch stream): median
        duration of each stage of the forward in ms, plus the plain (unmarked) time per bag of the same loop."""
        import numpy as np
        torch, L = self.torch, self._lib
        enc = self.enc
        x = self.bags[0]
        if self.mil is not None:
            x = torch.from_numpy(__import__("rrt_mil_amd").synth.bag(self.n, DIM, tag="bench/iso")).to(self.dev)
            need = enc._workspace(self.n, self.dev).numel()
            ws, y = torch.empty(need, dtype=torch.uint8, device=self.dev), torch.empty_like(x)
        else:
            ws, y = self.wss[0], self.outs[0]
        marks = [L.EV_START, L.EV_LN_PARTITION, L.EV_ATTN, L.EV_PROJ, L.EV_CR_COMBINE, L.EV_CR_INNER, L.EV_END]
        sets = [[self.hev.create() for _ in marks] for _ in range(reps)]

        def fwd(evs):
            self._lib.check(self.lib.rrt_encoder_forward_events_f32(C.byref(enc._desc), C.byref(self.w), x.data_ptr(),
                                                                    y.data_ptr(), self.n, ws.data_ptr(), ws.numel(),
                                                                    self.streams[0], evs), "forward")
        enc._desc.weights16_valid = 0
        solo_was, enc._desc.solo = enc._desc.solo, 1      # this pass IS one bag in flight
        fwd(None)
        enc._desc.weights16_valid = int(enc._desc.compute != L.COMPUTE_F32)
        lead = 100
        for i in range(lead + 4 * reps):
            evs = None
            if i >= lead and (i - lead) % 4 == 0:
                for j in range(L.EV_COUNT):
                    self.ev_arr[j] = None
                for m, e in zip(marks, sets[(i - lead) // 4]):
                    self.ev_arr[m] = e
                evs = self.ev_arr
            fwd(evs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(100):
            fwd(None)
        torch.cuda.synchronize()
        plain_ms = (time.perf_counter() - t0) / 100 * 1e3
        enc._desc.weights16_valid = 0
        enc._desc.solo = solo_was
        d = np.array([[self.hev.elapsed_ms(a, b) for a, b in zip(st[:-1], st[1:])] for st in sets])
        return dict(zip(("ln_partition", "fused_rmsa", "out_projection", "crmsa_combine", "crmsa_inner", "dispatch_ln"),
                        np.median(d, axis=0).tolist())), plain_ms

    def kernel_table(self, args):
        """roofline_kernels: each stage of a bag's forward against ITS roofline (north_star: MFMA utilisation for the dense
        projections, rocprof HBM GB/s for the streaming kernels)."""
        from rrt_mil_amd.geometry import region_grid
        stages, plain_ms = self.staged_pass()
        g, g8 = region_grid(self.n, self.enc_cfg["region_num"]), region_grid(self.n, 8)
        k, D, N = self.enc_cfg["crmsa_k"], DIM, self.n
        lowp = self.dtype in ("bf16", "f16")
        mpeak = PEAK_TFLOPS["bf16" if lowp else "f32"]
        es = 2 if lowp else 4                                   # bytes of a u / O element in HBM
        sc = 1 if self.enc_cfg.get("all_shortcut") else 0
        merged = bool(plan_flags(self.enc, self.n) & self._lib.PLAN_FUSED_PROJ)
        f_fused, _ = fused_flops(self.n, self.enc_cfg, with_proj=merged)
        tab = traffic_table(args.config, self.dtype)
        fused_pat = RMSA_CORE_KERNELS
        if merged:      # one launch: the two event marks behind it are a marker gap, counted with the launch
            stages["fused_rmsa"] += stages.pop("out_projection")
        spec = [
            ("ln_partition", "LayerNorm + zero-pad + region partition (rrt.py:121, rmsa.py:199-200,28-39)", "hbm",
             N * D * 4 + g.Np * D * es, ("ln_partition",)),
            ("fused_rmsa", "qkv projection + EPEG + softmax(QK^T)V per (region, head) (rmsa.py:100-122)"
             + (" + proj Linear + region_reverse + un-pad + residual as a later phase of the same launch (rmsa.py:131,41-54; "
                "rrt.py:125)" if merged else ""), "mfma", f_fused, fused_pat),
        ] + ([] if merged else [
            ("out_projection", "proj Linear + region_reverse + un-pad + residual (rmsa.py:131,41-54; rrt.py:125)", "mfma",
             2.0 * g.Np * D * D, ("linear_ws_kernel<6, 1, 1", "linear_ws_kernel<8, 1, 1", "linear_ws_kernel<9, 1, 1")),
        ]) + [
            ("crmsa_combine", "LN2 + logits + region softmax / min-max + combine (rmsa.py:303-316): x1 read once", "hbm",
             N * D * 4, ("crmsa_region4", "crmsa_logits", "crmsa_combine")),
            ("crmsa_inner", "MSA over the 64 k representatives: qkv, 64 x 64 attention, proj (rmsa.py:322)", "mfma",
             8.0 * k * 64 * D * D + 4.0 * k * 64 * 64 * D, ("linear_kernel<2", "region_attn64", "rmsa_fused16_kernel<4", "linear_ws_kernel<2")),
            ("dispatch_ln", "dispatch + residual (+ shortcut) + final LayerNorm (rmsa.py:324-335; rrt.py:192-195)", "hbm",
             N * D * 4 * (2 + sc), ("crmsa_dispatch_ln",)),
        ]
        out = []
        for key, what, bound, work, pats in spec:
            ms = stages[key]
            if bound == "mfma":
                ach, peak, unit = work / (ms * 1e-3) / 1e12, mpeak, "TFLOP/s"
            else:
                ach, peak, unit = work / (ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
            tr = traffic_of(tab, pats, per_forward_of=fused_pat)
            out.append({"stage": key, "kernel": what, "bound": bound, "achieved": round(ach, 2), "peak": peak, "unit": unit,
                        "frac": round(ach / peak, 4), "avg_ms": round(ms, 5),
                        "algorithmic": work, "traffic": tr["bytes"] if tr else None,
                        "traffic_kernels": tr["kernels"] if tr else None})
        note = ("one bag in flight, stage boundaries = the library's event marks (7 markers per instrumented forward, every "
                f"fourth forward, median of 12); the same loop without markers takes {plain_ms:.4f} ms per bag, the marked "
                f"stages add up to {sum(stages.values()):.4f} ms -- the difference is what the marker packets cost; "
                f"`traffic` = HBM-side bytes per forward of the stage's kernels ({os.path.relpath(TRAFFIC_FILE, ROOT)})")
        return out, note, plain_ms

    def finish(self, args, world, rank, elapsed):
        import numpy as np
        torch = self.torch
        for o in (self.call_outs if self.via == "forward_bags" else self.outs):
            assert torch.isfinite(o).all()
        if rank != 0:
            return {}
        rec = {}
        raw_elapsed = elapsed
        if self.via == "forward_bags":
            # the rounds 1-5 region, untimed, after the contract's: the C-ABI entry point per bag on S streams (S x Rr bags per
            # step), with the event marks around the dominant kernel -- printed as `raw_c_abi_loop` beside `value`
            self.enc._desc.compute, self.enc._desc.solo = self.compute, int(self.S == 1)
            for i in range(5):
                self.step_raw(i, False)
            self.sync()
            t0 = time.perf_counter()
            for i in range(self.raw_steps):
                self.step_raw(i, True)
            self.sync()
            raw_elapsed = time.perf_counter() - t0
            for o in self.outs_raw:
                assert torch.isfinite(o).all()
        merged = bool(plan_flags(self.enc, self.n) & self._lib.PLAN_FUSED_PROJ)
        flops, g = fused_flops(self.n, self.enc_cfg, with_proj=merged)
        peak = PEAK_TFLOPS["bf16" if self.dtype in ("bf16", "f16") else "f32"]
        if self.dtype == "f32x3":
            # projection: 3 bf16 MFMAs per product; attention: fp32 MFMA.  The roofline of THIS arithmetic is the sum of
            # the two parts' matrix-pipe times; `peak` = algorithmic FLOPs over that time
            f_proj, f_attn = 2.0 * g.Np * 1536 * DIM, 4.0 * g.Np * g.P * DIM
            peak = round(flops / (3.0 * f_proj / PEAK_TFLOPS["bf16"] + f_attn / PEAK_TFLOPS["f32"]), 1)
        iso_ms = self.isolated_fused_ms()
        tr = traffic_of(traffic_table(args.config, self.dtype), RMSA_CORE_KERNELS)
        kernel = (f"rmsa_fused_kernel (R-MSA per (region, head): qkv projection {g.P}x192x512 + EPEG + softmax(QK^T)V "
                  f"from LDS; {2.0 * g.Np * 1536 * DIM / 1e9:.2f} + {4.0 * g.Np * g.P * DIM / 1e9:.2f} GFLOP"
                  + (f"; then, as a later phase of the same launch's blocks, the out-projection {g.P}x64x512 slab of a region "
                     f"that finished a round earlier + un-partition + residual: {2.0 * g.Np * DIM * DIM / 1e9:.2f} GFLOP" if merged else "")
                  + f"), {self.dtype} operands"
                  + (" (projection: fp32 emulated by 3 bf16 MFMAs per product; attention: fp32 MFMA; peak = FLOPs over "
                     "3 x projection / 2.5 PFLOP/s + attention / 157.3 TFLOP/s)" if self.dtype == "f32x3" else ""))
        if self.mil is None:
            # all events against the window's first one: [start, end] of every launch on one clock
            ref = self.ev_pairs[0][0]
            iv = sorted((self.hev.elapsed_ms(ref, a), self.hev.elapsed_ms(ref, b)) for a, b in self.ev_pairs)
            raw_ms = float(np.mean([e - s for s, e in iv]))
            busy, cur_s, cur_e = 0.0, iv[0][0], iv[0][1]
            for s_, e_ in iv[1:]:
                if s_ > cur_e:
                    busy += cur_e - cur_s
                    cur_s, cur_e = s_, e_
                else:
                    cur_e = max(cur_e, e_)
            busy += cur_e - cur_s
            busy_ms = busy / len(iv)       # time the kernel was running at all, per launch
            # What the line itself allows: the step's S x R launches cannot have been running for longer than the step
            # took.  The instrumented steps carry two marker packets per forward and run slower than the steps around them
            # (round 4: one 16-launch step read 0.27-0.35 ms run to run) -- when the union figure contradicts ms_per_step it
            # is not evidence, and the line falls back to the bound that follows from ms_per_step alone.
            ms_per_step = raw_elapsed / self.raw_steps * 1e3          # of the raw loop (= the timed region under --raw-loop)
            per_step = self.S * self.Rr
            lb_ach = flops * per_step / (ms_per_step * 1e-3) / 1e12
            consistent = busy_ms * per_step <= ms_per_step
            ach = flops / (busy_ms * 1e-3) / 1e12 if consistent else lb_ach
            if self.via == "forward_bags":
                # `frac` of the line = what the line's own ms_per_step allows for the dominant kernel: its FLOPs x the step's
                # launches / ms_per_step / peak (every other kernel of the step counted as if it were this one's time)
                ms_line = elapsed / args.steps * 1e3
                ach = flops * self.S * self.R / (ms_line * 1e-3) / 1e12
            rp = rocprof_record(args.config, self.dtype, self.S)
            rec["roofline"] = {"bound": "mfma", "kernel": kernel, "achieved": round(ach, 2), "peak": peak,
                               "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               "frac_source": ("lower_bound_from_ms_per_step (timed region: RRTEncoder.forward_bags)" if self.via == "forward_bags"
                                               else "hip_event_union" if consistent else "lower_bound_from_ms_per_step"),
                               "frac_lower_bound": round((ach if self.via == "forward_bags" else lb_ach) / peak, 4),
                               "frac_raw_loop_lower_bound": round(lb_ach / peak, 4),
                               "frac_event_union": round(flops / (busy_ms * 1e-3) / 1e12 / peak, 4),
                               "flops_per_launch": flops,
                               "avg_launch_ms": round(raw_ms, 5), "busy_ms_per_launch": round(busy_ms, 5),
                               "launches": len(iv), "launches_per_step": self.S * self.R, "bags_in_flight": self.S,
                               "traffic": tr["bytes"] if tr else None, "traffic_source": tr["source"] if tr else None,
                               "rocprof": rp,
                               "note": ("`frac` = flops_per_launch x launches_per_step / ms_per_step / peak of THIS line's timed region "
                                        "(RRTEncoder.forward_bags, no event marks: the executor takes none), i.e. the whole step charged to the "
                                        "dominant kernel.  The event fields come from the raw C-ABI loop run after it (`raw_c_abi_loop`): "
                                        if self.via == "forward_bags" else "") +
                                       f"{len(iv)} launches = {self.ev_win} consecutive step(s) x {self.Rr} bags x {self.S} stream(s) in the middle of "
                                       "the raw loop, one HIP event pair per launch (recorded by librrt_hip on the launch stream).  "
                                       "avg_launch_ms = mean [start, end] interval of a launch (what rocprofv3's average duration of the "
                                       "kernel says; with several bags in flight the launches overlap and time-slice the matrix cores).  "
                                       "busy_ms_per_launch = the UNION of the launches' intervals / launches.  frac_event_union = "
                                       "flops_per_launch / busy_ms_per_launch / peak (the marker packets slow the instrumented steps: "
                                       "it is evidence only when busy x launches <= the raw loop's ms_per_step); "
                                       "frac_raw_loop_lower_bound = the ms_per_step bound of the raw loop.  `rocprof` = mean duration and "
                                       "union per launch from the committed rocprofv3 --kernel-trace table (tools/rocprof_union.py)"}
            if self.via == "forward_bags":
                rec["raw_c_abi_loop"] = {"value": round(self.S * self.Rr * self.raw_steps / raw_elapsed, 2), "unit": "slides/s",
                                         "steps": self.raw_steps, "bags_per_step": self.S * self.Rr,
                                         "ms_per_step": round(raw_elapsed / self.raw_steps * 1e3, 4),
                                         "note": "the timed region of rounds 1-5, untimed here (rank-local clock, after the contract's "
                                                 "region): rrt_encoder_forward_gated_f32 called per bag on S streams, no executor, "
                                                 "no fork / join; `value` / this = what the drop-in's batch call keeps of the raw rate"}
        ach = flops / (iso_ms * 1e-3) / 1e12
        iso = {"bound": "mfma", "kernel": kernel, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
               "frac": round(ach / peak, 4), "flops_per_launch": flops, "avg_launch_ms": round(iso_ms, 5),
               "traffic": tr["bytes"] if tr else None, "traffic_source": tr["source"] if tr else None,
               "note": "same kernel, untimed pass with one bag in flight (forwards back to back on one stream after 150 "
                       "forwards of lead, every fourth one instrumented, median of 10 launches)"}
        if self.mil is None:
            rec["roofline_isolated"] = iso
        else:
            rec["roofline"] = iso
        rec["roofline_kernels"], rec["roofline_kernels_note"], plain_ms = self.kernel_table(args)
        rec["one_bag_in_flight"] = {"ms_per_bag": round(plain_ms, 5), "slides_per_s": round(1e3 / plain_ms, 1),
                                    "note": "encoder forwards back to back on ONE stream (no second bag to fill the gaps)"}
        return rec


class MixWorkload:
    """config 4: one batch of mixed-size bags, LPT-split over the ranks, executor on every rank."""

    def __init__(self, args, rank, world, dev):
        import torch
        from rrt_mil_amd import RRTEncoder, _lib, sharding, synth
        self.torch = torch
        cfg = self.cfg = CONFIGS[args.config]
        self.enc_cfg = cfg["enc"]
        self.dtype = args.dtype or cfg["dtype"]
        self.S = max(1, args.streams)
        self.sizes = mix_sizes(cfg["n_bags"])
        e = self.enc_cfg
        self.assign = sharding.assign_bags(self.sizes, world, region_num=e["region_num"], epeg_k=e["epeg_k"],
                                           crmsa_k=e["crmsa_k"])
        self.mine = self.assign[rank]
        self.units_global = len(self.sizes)
        self.scaling = "strong"
        state = synth.encoder_state(**{k: v for k, v in e.items() if k != "region_num"})
        enc = RRTEncoder(**e).eval()
        enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()}, strict=True)
        self.enc = enc.to(dev)
        self.enc.compute_dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16}[self.dtype]
        big = torch.from_numpy(synth.bag(15000, DIM, tag=f"bench/mix/r{rank}")).to(dev)
        self.bags = [big[:self.sizes[i]].contiguous() for i in self.mine]
        self.outs = [torch.empty_like(b) for b in self.bags]
        self.caller = torch.cuda.Stream(dev) if dev.type == "cuda" else None
        loads = [sum(sharding.bag_cost(self.sizes[i], region_num=e["region_num"], epeg_k=e["epeg_k"],
                                       crmsa_k=e["crmsa_k"]) for i in r) for r in self.assign]
        self.extra = {"bags_per_rank": [len(r) for r in self.assign],
                      "cost_imbalance": round(max(loads) / (sum(loads) / len(loads)), 4),
                      "tokens_per_step": int(sum(self.sizes))}

    def step(self, i, timed):
        if self.bags:
            # (round 6) from a caller under its own torch stream: the call is asynchronous, the host prepares the next
            # call while the GPU runs this one -- on the default stream a 64-bag call blocks the host
            with self.torch.no_grad(), self.torch.cuda.stream(self.caller):
                self.enc.forward_bags(self.bags, streams=self.S, outs=self.outs)

    def sync(self):
        self.torch.cuda.synchronize()

    def finish(self, args, world, rank, elapsed):
        for o in self.outs:
            assert self.torch.isfinite(o).all()
        if rank != 0 or not self.bags:
            return {}
        # the R-MSA core kernel of every bag of this rank, one bag in flight, untimed: sum of the algorithmic FLOPs over the
        # sum of the launches' durations (HIP events recorded by the library around the kernel)
        from rrt_mil_amd import _lib
        torch, enc = self.torch, self.enc
        lib, hev = _lib.load(), HipEvents()
        enc._desc.compute = enc._compute_mode()
        w = enc._weights()
        dev = self.bags[0].device
        ws = enc._workspace(max(b.size(0) for b in self.bags), dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        ev_arr = (C.c_void_p * _lib.EV_COUNT)()
        pairs = []
        enc._desc.weights16_valid = 0
        for rep in range(2):                                        # first round warms up; the second is read
            pairs = []
            for b, o in zip(self.bags, self.outs):
                a_, b_ = hev.create(), hev.create()
                for j in range(_lib.EV_COUNT):
                    ev_arr[j] = None
                ev_arr[_lib.EV_LN_PARTITION], ev_arr[_lib.EV_ATTN] = a_, b_
                _lib.check(lib.rrt_encoder_forward_events_f32(C.byref(enc._desc), C.byref(w), b.data_ptr(), o.data_ptr(),
                                                              b.size(0), ws.data_ptr(), ws.numel(), st, ev_arr), "forward")
                enc._desc.weights16_valid = int(enc._desc.compute != _lib.COMPUTE_F32)
                pairs.append((a_, b_))
            torch.cuda.synchronize()
        enc._desc.weights16_valid = 0
        ms = sum(hev.elapsed_ms(a_, b_) for a_, b_ in pairs)
        flops = sum(fused_flops(b.size(0), self.enc_cfg)[0] for b in self.bags)
        peak = PEAK_TFLOPS["bf16" if self.dtype in ("bf16", "f16") else "f32"]
        ach = flops / (ms * 1e-3) / 1e12
        tab = traffic_table(args.config, self.dtype)
        tr = None
        if tab:
            hits = [v for k, v in tab.items() if any(p in k for p in RMSA_CORE_KERNELS)]
            nd = sum(v["dispatches"] for v in hits)
            if nd:
                tr = int(sum(v["hbm_bytes"] * v["dispatches"] for v in hits) / nd)
        return {"roofline": {"bound": "mfma", "kernel": "the R-MSA core kernel of each bag (rmsa_pair16 / rmsa_fused16 / rmsa_fused by "
                             f"region size): qkv projection + EPEG + softmax(QK^T)V per (region, head), {self.dtype} operands",
                             "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                             "flops_per_launch": flops / len(self.bags), "avg_launch_ms": round(ms / len(self.bags), 5),
                             "traffic": tr, "traffic_source": os.path.relpath(TRAFFIC_FILE, ROOT) if tr else None,
                             "note": f"sum over this rank's {len(self.bags)} bags (mixed sizes), one bag in flight, untimed pass "
                                     "after the timed region: sum of FLOPs / sum of the launches' durations; flops_per_launch, "
                                     "avg_launch_ms and traffic are means per launch"}}


# ------------------------------------------------------------------------------------ the rank program
def stabilise(wl, steps_hint, max_s=3.0, min_s=0.4):
    """Untimed: run probes of a few steps until three consecutive probe rates agree within 3 % (clock ramp,
    first-touch of the workspaces, HIP's launch pipeline) -- so that a short timed region (--steps 20) gives
    the same rate as a long one -- and for at least min_s seconds: the last 0.5-0.7 % of the ramp is a drift too slow for
    the 3 % test (round 5: `--steps 20 --warmup 5` read 5337 and then 5350, 5352, 5361 in its repeats, the 200-step default
    5377).  Bounded by max_s seconds.  Returns the number of steps spent."""
    probe = max(4, min(20, steps_hint))
    rates, spent = [], 0
    t_begin = time.perf_counter()
    t_end = t_begin + max_s
    while time.perf_counter() < t_end:
        wl.sync()
        t0 = time.perf_counter()
        for i in range(probe):
            wl.step(i, False)
        wl.sync()
        rates.append(probe / (time.perf_counter() - t0))
        spent += probe
        if len(rates) >= 3 and max(rates[-3:]) / min(rates[-3:]) < 1.03 and time.perf_counter() - t_begin >= min_s:
            break
    return spent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=1, choices=sorted(CONFIGS),
                    help="index into BASELINE.json configs (default 1: the config the metric is quoted on)")
    ap.add_argument("--dtype", choices=("f32", "bf16", "f16", "f32x3"), default=None,
                    help="override the config's arithmetic (f32x3: the two big projections emulated in fp32 on the bf16 "
                         "matrix cores, RRT_COMPUTE_F32X3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the informational extra records of the default run")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("RRT_BENCH_STREAMS", "0")),
                    help="bags in flight per GPU (one HIP stream + workspace each); a step = this many bags.  Default: 4 "
                         "(3 for bf16 / fp16 bags of > 12 k tokens and the configs[4] mix: measured optima, see main())")
    ap.add_argument("--bags-per-stream", type=int, default=0,
                    help="forwards per stream and step (configs 0-3; default 4): a step = streams x this many bags")
    ap.add_argument("--fresh-inputs", action="store_true",
                    help="configs 0-3: rotate over >= 512 MB of DISTINCT device-resident bags, each with its own output buffer "
                         "(more than the 256 MB Infinity Cache holds), instead of cycling four bags: what a loader that hands "
                         "over a new bag per iteration gives (main.py:434)")
    ap.add_argument("--raw-loop", action="store_true",
                    help="configs 0, 1, 3: time the C-ABI entry point called per bag on --streams streams (the timed region of "
                         "rounds 1-5) instead of RRTEncoder.forward_bags")
    ap.add_argument("--stub-cpu", action="store_true", help="rank logic only: CPU stand-in workload over gloo (tests)")
    ap.add_argument("--module-call-only", action="store_true",
                    help="print only the `module_call` record of this config (the default run spawns this as a child process)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args.gpus)          # does not return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE)")

    import torch
    from rrt_mil_amd import sharding
    dist = None
    if args.stub_cpu:
        dev = torch.device("cpu")
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
    else:
        assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # under torch.distributed.run, also at 1 rank
            import torch.distributed as dist
            dist.init_process_group("nccl", device_id=dev)   # RCCL

    cfg = CONFIGS[args.config]
    if args.streams <= 0:
        # fp32: measured on MI355X (round 4, profiles/r04_final_streams_sweep.txt): 2 -> 4.84 k, 3 -> 4.90 k, 4 -> 5.10 k, 5 -> 4.78 k,
        # 6 -> 4.85 k, 8 -> 5.07 k slides/s.  Kernels of different bags do not overlap on this chip in any way that saves
        # time (tools/corun_matrix.py: a kernel beside the fused R-MSA launch costs that launch the kernel's own solo
        # duration) -- what more bags in flight buy is the latency-bound CR-MSA chains of DIFFERENT bags running next to
        # each other; four streams = one per hardware pipe
        # bf16: 3 (17.8 k; 4: 17.7 k, 5: 14.4 k); the bf16 classifier of configs[2] (a longer chain per bag): 4 (9.7 k vs 9.1 k)
        # round 5 (profiles/r05_final_streams_sweep.txt): bf16 N = 9000 4 -> 19.05 k (3: 18.9 k); N = 30000 and the configs[4] mix keep 3
        # round 6 (same box, 2 / 3 / 4 bags in flight): configs[3] 5.90 / 6.02 / 5.97 k -> 3 stays; configs[4] 14.3 / 15.6 / 15.85 k -> 4
        lowp = (args.dtype or cfg["dtype"]) in ("bf16", "f16")
        args.streams = cfg.get("streams") or (3 if lowp and (cfg["n"] is None or cfg["n"] > 12000) else 4)
    if args.module_call_only and not args.stub_cpu and cfg["kind"] == "encoder":
        # a user's process: the module, its bags, nothing else (no bench streams / workspaces / event pools, whose streams
        # would take hardware queues in front of the executor's)
        from rrt_mil_amd import RRTEncoder, _lib, synth

        class _Bare:
            pass
        wl = _Bare()
        enc = RRTEncoder(**cfg["enc"]).eval()
        st = synth.encoder_state(**{k: v for k, v in cfg["enc"].items() if k != "region_num"})
        enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in st.items()}, strict=True)
        wl.enc, wl.S, wl.n = enc.to(dev), args.streams, cfg["n"]
        wl.dtype = args.dtype or cfg["dtype"]
        wl.compute = {"f32": _lib.COMPUTE_F32, "bf16": _lib.COMPUTE_BF16, "f16": _lib.COMPUTE_F16, "f32x3": _lib.COMPUTE_F32X3}[wl.dtype]
        wl.bags = [torch.from_numpy(synth.bag(wl.n, DIM, tag=f"bench/r{rank}/b{i}")).to(dev) for i in range(4)]
        var = os.environ.get("RRT_BENCH_MC_VARIANTS")
        print(json.dumps(module_call(wl, dev, variants=tuple(var.split(",")) if var else ("loop", "default", "async"))), flush=True)
        return
    if args.stub_cpu:
        wl = StubWorkload(args, rank, world, dev)
    elif cfg["kind"] == "mix":
        wl = MixWorkload(args, rank, world, dev)
    else:
        wl = EncoderWorkload(args, rank, world, dev)

    for i in range(args.warmup):
        wl.step(i, False)
    wl.sync()
    ramp = stabilise(wl, args.steps) if not args.stub_cpu else 0
    if dist:
        dist.barrier()
    wl.sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        wl.step(i, True)
    wl.sync()
    if dist:
        dist.barrier()
    wl.sync()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, device=None if args.stub_cpu else dev)   # whole-job time = slowest rank

    # run-to-run spread: three more repeats of the same K steps, same bracketing, after the contract's timed region
    # (rank-local clocks, informational: `value` is the first region, the only one bracketed by the barriers' MAX)
    spread = []
    if not args.stub_cpu:
        for _ in range(3):
            wl.sync()
            ts = time.perf_counter()
            for i in range(args.steps):
                wl.step(i, False)
            wl.sync()
            spread.append(wl.units_global * args.steps / (time.perf_counter() - ts))

    rec_extra = wl.finish(args, world, rank, elapsed)

    if rank == 0:
        dtype = "f32" if args.stub_cpu else wl.dtype
        ms_per_step = elapsed / args.steps * 1e3
        value = wl.units_global * args.steps / elapsed
        enc_cfg = cfg["enc"]
        backend = None
        if dist:
            backend = str(dist.get_backend())
        config = {"workload": cfg["label"] + (" [--stub-cpu: CPU stand-in, rank logic only]" if args.stub_cpu else ""),
                  "collectives": {"backend": backend, "world": world,
                                  "use": "barrier + MAX all-reduce of the elapsed time only (no data-path collective)"},
                  "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                  "baseline_config_index": args.config, "dim": DIM, "bags_per_step": wl.units_global,
                  "streams_per_gpu": getattr(wl, "S", None), "bags_per_stream_per_step": getattr(wl, "R", None),
                  "timed_region_via": {"forward_bags": "rrt_mil_amd.RRTEncoder.forward_bags (one call per step, caller under its own torch stream)",
                                       "raw_loop": "C ABI per bag (rrt_encoder_forward_gated_f32 / rrt_mil_forward_f32)"}.get(getattr(wl, "via", None),
                                                                                                               "RRTEncoder.forward_bags" if cfg["kind"] == "mix" else None),
                  "untimed_ramp_steps": ramp,
                  "parallelism": f"bag-parallel x{world} (no data-path collective)"}
        if cfg["n"]:
            config["n_tokens"] = cfg["n"]
            extra_f = 2.0 * cfg["n"] * cfg.get("input_dim", 0) * DIM
            gf = flops_total(cfg["n"], enc_cfg) + extra_f
            config["gflop_per_bag"] = round(gf / 1e9, 2)
            config["whole_path_tflops"] = round(wl.units_global / world * gf / (ms_per_step * 1e-3) / 1e12, 2)
            wp_f, wp_b = gf, bytes_total(cfg["n"], cfg.get("input_dim", 0))
        else:
            sizes = mix_sizes(cfg["n_bags"])
            gf = sum(flops_total(n, enc_cfg) for n in sizes)
            config["gflop_per_step"] = round(gf / 1e9, 2)
            config["whole_path_tflops"] = round(gf / (ms_per_step * 1e-3) / 1e12 / world, 2)
            wp_f, wp_b = gf / len(sizes), sum(bytes_total(n) for n in sizes) / len(sizes)     # per average bag of the mix
        whole_path = None
        if not args.stub_cpu:
            whole_path = whole_path_roofline(value / world, wp_f, wp_b, dtype, traffic_per_forward(args.config, dtype, cfg["kind"]))
        config.update(wl.extra)
        metric = ("slides/sec RRTEncoder fwd, N=9000 D=512 region_num=8" if args.config == 1
                  else f"slides/sec, BASELINE configs[{args.config}]")
        rec = {"metric": metric, "value": round(value, 2), "unit": "slides/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
               "scaling": wl.scaling, "vs_baseline": None, "dtype": dtype,
               "data": "stub" if args.stub_cpu else "synthetic", "config": config}
        if spread:
            allv = [value] + spread
            rec["value_spread"] = {"values": [round(v, 1) for v in allv], "min": round(min(allv), 1), "max": round(max(allv), 1),
                                   "rel": round((max(allv) - min(allv)) / value, 4),
                                   "note": f"the timed region's value followed by 3 repeats of the same {args.steps} steps "
                                           "(this rank's clock, no barrier between them)"}
        rec.update(rec_extra)
        if whole_path is not None:
            rec["whole_path"] = whole_path
        if not args.stub_cpu and world == 1 and cfg["kind"] == "encoder" and not args.no_extras:
            rec["module_call"] = module_call_record(args)
            if args.config == 1 and not args.dtype:      # the headline config also through the module in bf16 (the --amp path)
                import copy
                a16 = copy.copy(args)
                a16.dtype = "bf16"
                rec["module_call_bf16"] = module_call_record(a16)
        if not args.stub_cpu and args.config == 1 and world == 1 and not args.no_extras:
            rec.update(extras(wl, dev))
            for c in (0, 2, 3, 4):
                rec[f"config{c}"] = config_record(c)
            rec["fresh_inputs"] = fresh_inputs_record(rec)
        if not args.stub_cpu and world == 1 and not args.no_cpu_baseline:
            n_cpu = cfg["n"] or 9000          # config 4: the typical bag of the mix
            rec["cpu_baseline"] = cpu_baseline(n_cpu, enc_cfg)
        print(json.dumps(rec), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def h2d_inclusive(wl, dev, n_bags=96):
    """slides/s when every bag starts in HOST memory: (a) pinned host bags through rrt_mil_amd.BagFeeder (copy stream +
    events, the H2D of the next bags under the current forward), (b) pageable host bags through the feeder, (c) the
    reference's loop: blocking bag.to(device) then forward.  One stream of forwards (the copies are the other stream)."""
    import torch
    from rrt_mil_amd import BagFeeder
    enc, n = wl.enc, wl.n
    host = [torch.randn(n, DIM) for _ in range(8)]
    pinned = [h.pin_memory() for h in host]
    y = torch.empty(n, DIM, device=dev)
    res = {}
    with torch.no_grad():
        for name, src in (("pinned_feeder", pinned), ("pageable_feeder", host)):
            for rep in range(2):                               # first pass warms the staging buffers / allocator
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for xb in BagFeeder((src[i % len(src)] for i in range(n_bags)), device=dev, depth=3):
                    enc.forward_bag(xb, out=y)
                torch.cuda.synchronize()
                res[name] = n_bags / (time.perf_counter() - t0)
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_bags):
                enc.forward_bag(host[i % len(host)].to(dev), out=y)      # the reference's per-iteration blocking copy
            torch.cuda.synchronize()
            res["blocking_to_device"] = n_bags / (time.perf_counter() - t0)
    gb = n * DIM * 4 / 1e9
    return {"unit": "slides/s", "pinned_feeder": round(res["pinned_feeder"], 1),
            "pageable_feeder": round(res["pageable_feeder"], 1), "blocking_to_device": round(res["blocking_to_device"], 1),
            "pinned_feeder_h2d_GBps": round(res["pinned_feeder"] * gb, 1), "bag_mb": round(gb * 1e3, 1), "bags": n_bags,
            "note": "fp32 encoder, N=9000 D=512, one forward stream + the feeder's copy stream; PCIe Gen5 x16 moves ~53 GB/s, i.e. "
                    "~2.9 k of these bags per second: with host-resident bags the link, not the encoder, is the limit "
                    "(rrt_mil_amd/feed.py; reference loop: main.py:434)"}


def h2d_inclusive_c2(dev, n_bags=64, n=9000, in_dim=1024):
    """BASELINE configs[2] (the C16-R50 classifier under bf16 arithmetic) with every slide starting in PINNED HOST memory:
    features shipped in fp32 (36.9 MB per slide) and in bf16 (18.4 MB: rrt_mil_amd.BagFeeder(dtype=torch.bfloat16); the logits
    are bit-identical, tests/test_hip_parity.py::test_bag_feeder_16bit_features_bit_identical_logits)."""
    import torch
    from rrt_mil_amd import BagFeeder, RRTMIL, synth
    mil = RRTMIL(input_dim=in_dim, n_classes=2, epeg_k=15, crmsa_k=1, all_shortcut=True).eval()
    mst = synth.mil_state(input_dim=in_dim, n_classes=2, epeg_k=15, crmsa_k=1)
    mil.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in mst.items()}, strict=True)
    mil = mil.to(dev)
    mil.online_encoder.compute_dtype = torch.bfloat16
    host32 = [torch.randn(n, in_dim).relu_().pin_memory() for _ in range(6)]
    host16 = [h.to(torch.bfloat16).pin_memory() for h in host32]
    res = {}
    with torch.no_grad():
        for name, src, dt in (("fp32_features", host32, torch.float32), ("bf16_features", host16, torch.bfloat16)):
            for rep in range(2):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for xb in BagFeeder((src[i % len(src)] for i in range(n_bags)), device=dev, depth=3, dtype=dt):
                    lg = mil.forward_bag(xb)
                torch.cuda.synchronize()
                res[name] = n_bags / (time.perf_counter() - t0)
            assert torch.isfinite(lg).all()
    mb32 = n * in_dim * 4 / 1e6
    return {"unit": "slides/s", "fp32_features": round(res["fp32_features"], 1), "bf16_features": round(res["bf16_features"], 1),
            "fp32_h2d_GBps": round(res["fp32_features"] * mb32 / 1e3, 1), "bf16_h2d_GBps": round(res["bf16_features"] * mb32 / 2e3, 1),
            "slide_mb": {"fp32": round(mb32, 1), "bf16": round(mb32 / 2, 1)}, "slides": n_bags,
            "note": "RRTMIL (C16-R50 shape, bf16 arithmetic) fed from pinned host memory through BagFeeder, one forward stream + the "
                    "copy stream; device-resident the same classifier runs `config2.value` slides/s: with fp32 features the link "
                    "(~53 GB/s = ~1.4 k slides/s) is the limit, 16-bit features double it (rrt_mil_desc.input16; "
                    "modules/rrt.py:208-229, dataloader.py:181,198)"}


def module_call(wl, dev, n_bags=64, variants=("loop", "default", "async")):
    """The boundary the reference exposes is the nn.Module (modules/rrt.py:133-202), not the C ABI the timed region calls:
    the same bags through `enc(bag)` one at a time (the reference's loop, main.py:466-467; one bag in flight) and through
    `enc.forward_bags(bags, streams=S)` (S in flight), with the host's own time per forward (the Python + ctypes cost of one
    call, measured while the GPU queue is far from full, i.e. not waiting for the device)."""
    import torch
    enc, S = wl.enc, wl.S
    # the asynchronous caller's stream is created FIRST, before the executor creates the process's bag streams: HIP maps streams to
    # hardware queues in creation order, and a caller stream created after them (as rounds 5-6 did here) came to share a queue
    # with a bag stream -- its fork / join waits then serialise that stream's bags (4.5 k fp32 / 12.6 k bf16 where the timed
    # region, whose caller exists from the workload's construction on, gets 5.27 k / 19.4 k)
    caller = torch.cuda.Stream(dev)
    bags3 = [b.unsqueeze(0) for b in wl.bags]
    mode_was, solo_was = enc.compute_dtype, enc.__dict__.get("solo", True)
    enc.compute_dtype = {"f32": torch.float32, "bf16": torch.bfloat16, "f16": torch.float16, "f32x3": "f32x3"}[wl.dtype]
    out = {"unit": "slides/s", "dtype": wl.dtype, "n_tokens": wl.n}
    with torch.no_grad():
        enc.solo = True
        y = None
        for i in range(200):                          # (a fresh process: clocks ramp over the first tens of milliseconds)
            y = enc(bags3[i % len(bags3)])
        torch.cuda.synchronize()
        reps = []
        n_loop = 5 if "loop" in variants else 0
        for _ in range(n_loop):                       # 64 forwards are 14 ms: one host hiccup (round 5 saw 3.7 k among 4.5 k) would be the record
            t0 = time.perf_counter()
            for i in range(n_bags):
                y = enc(bags3[i % len(bags3)])
            host = time.perf_counter() - t0           # enqueue only: n_bags forwards are ~600 packets, the queue does not fill
            torch.cuda.synchronize()
            reps.append((time.perf_counter() - t0, host))
        if reps:
            reps.sort()
            t1, host = reps[len(reps) // 2]           # the median repeat
            out["module_loop"] = round(n_bags / t1, 1)
            out["module_loop_repeats"] = [round(n_bags / t, 1) for t, _ in reps]
            out["host_us_per_bag"] = round(host / n_bags * 1e6, 1)
        n_batch = 4 * n_bags                          # one executor call = one fork / join: the longer the batch, the less it weighs
        batch = [bags3[i % len(bags3)] for i in range(n_batch)]
        outs = [torch.empty_like(b[0]) for b in batch]
        # untimed calls over the whole batch first: they create the executor and TOUCH the 256 output buffers (4.7 GB of
        # fresh device memory: the first write to a new allocation pays for its page mapping -- round 5 measured 1.9-2.6 k
        # slides/s for a first call against 5.0-5.2 k from the second on, tools/bench_bags.py).  Then FIVE calls back to back
        # (rounds 4-5 timed ONE call that started on an idle, down-clocked GPU right after a device sync and read 4-15 % low:
        # 5.06 k fp32 / 16.6 k bf16 where five consecutive calls of the same process give 5.26 k / 19.3 k, profiles/r06_probe1.txt).
        # ONE caller stream per process (round 6, tools/experiments/r06_probe_modcall.py): whichever of the two variants ran second in
        # one process read 4.5 k fp32 / 12.9 k bf16 -- the executor carries the first share of a call's bags on the CALLER's
        # stream, so a second caller is a fifth hardware queue in use, and the chip schedules four.
        # module_call_record therefore measures the two in two child processes.
        for name, ctx in (("forward_bags_default_stream", contextlib.nullcontext()), ("forward_bags", torch.cuda.stream(caller))):
            if ("default" if name.endswith("default_stream") else "async") not in variants:
                continue
            with ctx:
                for _ in range(2):
                    enc.forward_bags(batch, streams=S, outs=outs)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    enc.forward_bags(batch, streams=S, outs=outs)
                host_b = time.perf_counter() - t0
                torch.cuda.synchronize()
                t1 = time.perf_counter() - t0
            out[name] = round(5 * n_batch / t1, 1)
            out[name + "_host_us_per_bag"] = round(host_b / (5 * n_batch) * 1e6, 1)
        out["forward_bags_streams"] = S
        out["forward_bags_batch"] = n_batch
        assert (y is None or torch.isfinite(y).all()) and torch.isfinite(outs[-1]).all()
    enc.compute_dtype, enc.solo = mode_was, solo_was
    enc._desc.compute = wl.compute
    out["note"] = (f"{n_bags} device-resident bags: `for bag in bags: enc(bag)` under no_grad through nn.Module.__call__ (one bag "
                   f"in flight, the reference's loop) and five consecutive enc.forward_bags({n_batch} bags, streams={S}) calls -- from the "
                   "process's default stream (`forward_bags_default_stream`: each call blocks the host until its bags are done) and from "
                   "a caller under `with torch.cuda.stream(s):` (`forward_bags`: asynchronous, what bench.py's timed region does); "
                   "host_us_per_bag = host thread time per forward (for the blocking call: its wall time); compare module_loop with "
                   "one_bag_in_flight.slides_per_s and forward_bags with `value`")
    return out


def module_call_record(args):
    """`module_call` of this config from a CHILD process (bench.py --module-call-only): what a user's process gets.  In the
    process that has just run the timed region -- its bag streams, event pools and torch's own streams already mapped to
    hardware queues in creation order -- the executor's streams share queues with them (measured in round 5: 4.48 k slides/s
    for forward_bags in-process against 5.1-5.2 k in a fresh process, tools/bench_bags.py)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--config", str(args.config), "--module-call-only", "--no-cpu-baseline"]
    if args.dtype:
        cmd += ["--dtype", args.dtype]
    r = None
    for var in ("loop,default", "async"):       # one caller stream per process (module_call's note)
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=dict(os.environ, RRT_BENCH_MC_VARIANTS=var))
            q = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception as e:                  # the headline line must not die with a side record
            return {"error": f"{type(e).__name__}: {e}"[:300]}
        if r is None:
            r = q
        else:
            r.update({k: v for k, v in q.items() if k.startswith("forward_bags") and not k.startswith("forward_bags_default")})
    r["command"] = "RRT_BENCH_MC_VARIANTS=loop,default | async  bench.py " + " ".join(cmd[2:])
    return r


def config_record(c):
    """A bounded single-GPU record of BASELINE configs[c] inside the default run (rank 0, after the timed region):
    `bench.py --config c` with a short timed region, run as a CHILD PROCESS while this one idles -- HIP maps streams to
    hardware queues in creation order, and a process that has already created this run's streams (four bags in flight, the
    feeder's copy stream, the executor's own) no longer gives a new workload the queues a fresh process gets (measured:
    configs[4] 7.8 k slides/s in-process vs 14.5 k in its own process)."""
    import subprocess
    steps = {0: 100, 2: 30, 3: 12, 4: 12}[c]        # (configs 0-3: a step = streams x 4 bags)
    t0 = time.perf_counter()
    cmd = [sys.executable, os.path.abspath(__file__), "--config", str(c), "--steps", str(steps), "--warmup", "5",
           "--no-extras", "--no-cpu-baseline"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        r = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as e:                      # the headline line must not die with a side record
        return {"error": f"{type(e).__name__}: {e}"[:300]}
    rec = {"value": r["value"], "unit": r["unit"], "dtype": r["dtype"], "n_gpus": 1, "steps": r["steps"],
           "ms_per_step": r["ms_per_step"], "streams_per_gpu": r["config"].get("streams_per_gpu"),
           "bags_per_step": r["config"].get("bags_per_step"), "workload": r["config"]["workload"],
           "whole_path_tflops": r["config"].get("whole_path_tflops"), "value_spread": (r.get("value_spread") or {}).get("values")}
    for key in ("roofline", "roofline_isolated"):
        if key in r:
            rec[key] = {k: r[key][k] for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "avg_launch_ms", "traffic")
                        if k in r[key]}
    if "roofline_kernels" in r:                 # every stage against its own bound: HBM GB/s for the streaming stages
        rec["stages"] = [{k: st[k] for k in ("stage", "bound", "achieved", "peak", "unit", "frac", "avg_ms", "traffic")}
                         for st in r["roofline_kernels"]]
    if "one_bag_in_flight" in r:
        rec["one_bag_in_flight_ms"] = r["one_bag_in_flight"]["ms_per_bag"]
    if "whole_path" in r:                       # both roofs + counted traffic over algorithmic bytes
        rec["whole_path"] = r["whole_path"]
    rec["command"] = "bench.py " + " ".join(cmd[2:])
    rec["wall_s"] = round(time.perf_counter() - t0, 1)
    return rec


def fresh_inputs_record(rec):
    """Review item 5 (round 5): the same lines with FRESH inputs -- every step's bags taken from >= 512 MB of distinct
    device-resident bags (twice the 256 MiB Infinity Cache) and written to as many distinct outputs, so that neither a bag
    nor its output is cache-resident when its forward starts (the reference moves one new bag per iteration, main.py:434).
    One child process per line; `cycled` = the value of this run's own record of the same configuration (4 bags, 2 at
    N = 30000, reused: MALL-resident)."""
    import subprocess
    out = {"note": "bench.py --fresh-inputs: >= 512 MB of distinct inputs in rotation (config 2: distinct 1024-wide feature bags), "
                   "one distinct output per bag; `cycled` = this run's record of the same configuration.  config 4's one batch IS "
                   "64 distinct bags (1.2 GB of inputs per step): fresh by construction, no second line.  The HBM-bound stage "
                   "fractions of `roofline_kernels` refer to the cycled inputs of the one-bag-in-flight pass (DESIGN.md section 5)"}
    cyc = {"config1_f32": rec.get("value"), "config1_bf16": (rec.get("amp_bf16") or {}).get("value"),
           "config2": (rec.get("config2") or {}).get("value"), "config3": (rec.get("config3") or {}).get("value")}
    for key, args, steps in (("config1_f32", ["--config", "1", "--dtype", "f32"], 6), ("config1_bf16", ["--config", "1", "--dtype", "bf16"], 8),
                             ("config2", ["--config", "2"], 40), ("config3", ["--config", "3"], 8)):
        row = {"cycled": cyc.get(key)}
        cmd = [sys.executable, os.path.abspath(__file__)] + args + ["--steps", str(steps), "--warmup", "3", "--no-extras",
                                                                  "--no-cpu-baseline", "--fresh-inputs"]
        try:
            r = json.loads(subprocess.run(cmd, capture_output=True, text=True, timeout=240).stdout.strip().splitlines()[-1])
            row["fresh"] = r["value"]
            row["distinct"] = r["config"].get("fresh_inputs")
        except Exception as e:
            row["fresh"] = f"{type(e).__name__}: {e}"[:200]
        if isinstance(row.get("fresh"), float) and isinstance(row.get("cycled"), float):
            row["fresh_over_cycled"] = round(row["fresh"] / row["cycled"], 4)
        out[key] = row
    return out


def extras(wl, dev):
    """Informational records of the default run (rank 0, after the timed region; none is the headline value)."""
    import torch
    from rrt_mil_amd import RRTEncoder, RRTMIL, _lib, synth
    out = {}
    S = wl.S
    def set_mode(mode):                       # the timed region's arithmetic: the C-ABI descriptor AND the module's compute_dtype
        wl.enc._desc.compute = mode
        wl.enc.compute_dtype = {_lib.COMPUTE_F32: torch.float32, _lib.COMPUTE_BF16: torch.bfloat16,
                                _lib.COMPUTE_F16: torch.float16, _lib.COMPUTE_F32X3: "f32x3"}[mode]
    out0 = (wl.call_outs if wl.via == "forward_bags" else wl.outs)
    # the same workload with bf16 arithmetic (the reference's --amp / autocast path, BASELINE configs[2..4])
    set_mode(_lib.COMPUTE_BF16)
    for i in range(5):
        wl.step(i, False)
    torch.cuda.synchronize()
    ta = time.perf_counter()
    for i in range(20):
        wl.step(i, False)
    torch.cuda.synchronize()
    out["amp_bf16"] = {"value": round(wl.units_global * 20 / (time.perf_counter() - ta), 2), "unit": "slides/s", "n_gpus": 1,
                       "note": "rank 0 only, 20 steps after the timed region, same path (RRTEncoder.forward_bags); RRT_COMPUTE_BF16 "
                               "(bf16 operands on the matrix cores, fp32 accumulate; `--config 3` / `--dtype bf16` give the full record)"}
    # fp32 EMULATED on the bf16 matrix cores for the two big projections (RRT_COMPUTE_F32X3; ~1e-6 from the exact path)
    set_mode(_lib.COMPUTE_F32X3)
    for i in range(5):
        wl.step(i, False)
    torch.cuda.synchronize()
    tb = time.perf_counter()
    for i in range(20):
        wl.step(i, False)
    torch.cuda.synchronize()
    tb = time.perf_counter() - tb
    y_x3 = out0[0].clone()
    set_mode(_lib.COMPUTE_F32)
    wl.step(19, False)                        # the same bags through the exact path: out0[0] of both modes side by side
    torch.cuda.synchronize()
    out["f32x3"] = {"value": round(wl.units_global * 20 / tb, 2), "unit": "slides/s", "n_gpus": 1,
                    "max_abs_vs_exact_f32": float((y_x3 - out0[0]).abs().max()),
                    "note": "rank 0 only, 20 steps after the timed region; RRT_COMPUTE_F32X3: qkv / proj GEMMs of the R-MSA "
                            "layers as three bf16 MFMAs per product on (hi, lo) bf16 operand pairs, fp32 accumulate; attention, "
                            "LayerNorm, CR-MSA exact fp32 (`--dtype f32x3` gives the full record)"}

    # PCIe-inclusive rates (the reference moves one bag per iteration with a blocking
    # bag.to(device), main.py:434): host bags -> HBM -> encoder, fp32, 18.4 MB per bag.  Never `value`.
    out["h2d_inclusive"] = h2d_inclusive(wl, dev)
    out["h2d_inclusive_c2"] = h2d_inclusive_c2(dev)

    # the whole slide classifier of BASELINE configs[2] (C16-R50 shape) through the one-call path (row f1), fp32,
    # one bag in flight
    mcfg = dict(input_dim=1024, n_classes=2, epeg_k=15, crmsa_k=1, all_shortcut=True)
    mil = RRTMIL(**mcfg).eval()
    mst = synth.mil_state(input_dim=1024, n_classes=2, epeg_k=15, crmsa_k=1)
    mil.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in mst.items()}, strict=True)
    mil = mil.to(dev)
    feats = torch.from_numpy(synth.bag(9000, 1024, tag="mil", nonneg=True)).to(dev).unsqueeze(0)
    with torch.no_grad():
        for _ in range(5):
            lg = mil(feats)
        torch.cuda.synchronize()
        tm = time.perf_counter()
        for _ in range(50):
            lg = mil(feats)
        torch.cuda.synchronize()
    tm = (time.perf_counter() - tm) / 50
    assert torch.isfinite(lg).all()
    out["rrtmil_c16"] = {"value": round(1.0 / tm, 2), "unit": "slides/s", "ms_per_slide": round(tm * 1e3, 4), "n_gpus": 1,
                         "note": "RRTMIL(input_dim=1024, epeg_k=15, crmsa_k=1, all_shortcut=True).eval() forward, "
                                 "N=9000 x 1024 -> logits, fp32, one bag in flight, rank 0 after the timed region "
                                 "(rrt_mil_forward_f32: patch_to_emb GEMM+ReLU, encoder, DAttention pooling, predictor)"}
    del mil, feats

    # one training step of the same encoder (row f2): forward with stash + full backward, fp32, default proj
    # dropout 0.1, one bag per step
    tenc = RRTEncoder(**wl.enc_cfg).to(dev).train()
    tenc.load_state_dict(wl.enc.state_dict())
    xg = wl.bags[0].unsqueeze(0)
    gy = torch.randn_like(xg)
    torch.cuda.reset_peak_memory_stats(dev)

    def tstep():
        tenc.zero_grad(set_to_none=True)
        tenc(xg).backward(gy)                 # (rounds 3-5 timed `(y * g).sum().backward()`: three torch kernels, 33 us, in the step)
    for _ in range(3):
        tstep()
    torch.cuda.synchronize()
    tt = time.perf_counter()
    for _ in range(20):
        tstep()
    torch.cuda.synchronize()
    tt = (time.perf_counter() - tt) / 20
    f_fwd = flops_total(wl.n, wl.enc_cfg)
    f_train = 3.0 * f_fwd                     # forward + (dX and dW of every product of the forward); the attention
    #                                           backward's recomputation of S in LDS is extra work, not counted
    out["train_step"] = {"ms_per_step": round(tt * 1e3, 4), "steps_per_s": round(1.0 / tt, 2),
                         "peak_mem_mb": round(torch.cuda.max_memory_allocated(dev) / 1e6, 1),
                         "roofline": {"bound": "mfma", "flops_per_step": f_train,
                                      "formula": "3 x F_forward (SURVEY 8(d) F; backward = dX + dW of each product; recomputed "
                                                 "scores not counted)",
                                      "achieved": round(f_train / tt / 1e12, 2), "peak": PEAK_TFLOPS["f32"], "unit": "TFLOP/s",
                                      "frac": round(f_train / tt / 1e12 / PEAK_TFLOPS["f32"], 4),
                                      "kernel_table": "profiles/r06_final_train_kernel_stats.txt (tools/prof_train.sh)"},
                         "note": "RRTEncoder.train() forward (stash) + backward of every parameter, N=9000 D=512, fp32, "
                                 "drop_out=0.1, one bag per step (rrt_encoder_forward_train_f32 / rrt_encoder_backward_f32)"}
    return out


if __name__ == "__main__":
    main()
