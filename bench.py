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
In configs 0-3 a step = `--streams` (default 2) independent bags per GPU, each an ordinary forward on its own HIP
stream with its own workspace (bags are independent units), and every rank owns its own bags
("scaling": "weak").  No data-path collective anywhere: RCCL carries the barrier and a MAX of the elapsed time.

Besides the contract fields the JSON line carries
  roofline     -- the dominant kernel (the fused R-MSA kernel): algorithmic FLOPs per launch / its average
                  duration, measured live with HIP events that librrt_hip records on the launch stream inside
                  the timed region (config 2: in an untimed pass of the same encoder; the one-call classifier
                  entry takes no event array);
  cpu_baseline -- the oracle's eager torch-CPU port of the reference op sequence
                  (oracle/rrt_oracle.py::forward_eager) timed on this box's host cores over a bounded sample
                  of the same workload (rank 0, N=1 only).
--dtype overrides a config's arithmetic (f32 | bf16 | f16 | f32x3 = fp32 in / out with the big products emulated on the bf16
matrix cores, RRT_COMPUTE_F32X3); the default run also reports bf16 and f32x3 as extra records (`amp_bf16`, `f32x3`).
--stub-cpu replaces the GPU workload by a tiny CPU one over gloo: the rank logic (spawn, sharding, barrier,
MAX over ranks, the JSON line) then runs without a GPU -- tests/test_multiproc_cpu.py drives it.
"""
import argparse
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
# HBM-side bytes of the dominant kernel per launch at the north star from rocprofv3 --pmc (separate FETCH_SIZE
# and WRITE_SIZE passes, FETCH_SIZE doubled per the gfx950 correction)
TRAFFIC_BYTES_PER_LAUNCH = {("f32", 9000): 88.0e6,      # profiles/r02_a_pmc_f32.txt  (2 x 33967.5 KiB + 18432 KiB)
                            ("bf16", 9000): 31.7e6,     # profiles/r02_a_pmc_bf16.txt (2 x 10876.6 KiB +  9216 KiB)
                            ("f32x3", 9000): 88.0e6}    # profiles/r02_b_pmc_f32x3.txt (2 x 33959.8 KiB + 18432 KiB)

CONFIGS = {
    0: dict(kind="encoder", n=512, dtype="f32", enc=dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8),
            label="BASELINE configs[0]: RRTEncoder(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8).eval() forward, "
                  "one device-resident bag N=512 D=512 per stream per step"),
    1: dict(kind="encoder", n=9000, dtype="f32", enc=dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8),
            label="BASELINE configs[1]: RRTEncoder(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8).eval() forward, "
                  "one device-resident bag N=9000 D=512 per GPU per step, fp32, closed-form weights"),
    2: dict(kind="mil", n=9000, dtype="bf16", input_dim=1024,
            enc=dict(mlp_dim=512, epeg_k=15, crmsa_k=1, region_num=8, all_shortcut=True),
            label="BASELINE configs[2]: C16-R50 RRTMIL(input_dim=1024, epeg_k=15, crmsa_k=1, all_shortcut=True).eval() "
                  "forward, N=9000 x 1024 non-negative features -> logits (rrt_mil_forward_f32: fc + ReLU, encoder, "
                  "DAttention pooling, predictor), bf16 autocast-class arithmetic"),
    3: dict(kind="encoder", n=30000, dtype="bf16", enc=dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=16),
            label="BASELINE configs[3]: survival long sequence, RRTEncoder(region_num=16, epeg_k=15, crmsa_k=3).eval() "
                  "forward, device-resident bags N=30000 D=512 owned by each GPU, bf16 autocast-class arithmetic"),
    4: dict(kind="mix", n=None, dtype="bf16", n_bags=64, enc=dict(mlp_dim=512, epeg_k=21, crmsa_k=5, region_num=8),
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


def fused_flops(n, enc_cfg):
    """The fused R-MSA kernel's share: qkv projection [Np, D] x [3D, D]^T + Q K^T + A V per (region, head)."""
    from rrt_mil_amd.geometry import region_grid
    g = region_grid(n, enc_cfg["region_num"])
    return 2.0 * g.Np * (3 * DIM) * DIM + 4.0 * g.Np * g.P * DIM, g


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
    rec = {"value": round(1.0 / med, 3), "unit": "slides/s", "cores": cores, "kind": "port",
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
        self.dev, self.n, self.enc_cfg = dev, cfg["n"], cfg["enc"]
        self.dtype = args.dtype or cfg["dtype"]
        self.compute = {"f32": _lib.COMPUTE_F32, "bf16": _lib.COMPUTE_BF16, "f16": _lib.COMPUTE_F16,
                        "f32x3": _lib.COMPUTE_F32X3}[self.dtype]
        self.S = S = max(1, args.streams)
        self.units_global = world * S
        self.scaling = "weak"
        self.lib = _lib.load()
        self.hev = HipEvents()
        tstreams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(S - 1)]
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
            self.mdesc, self.mw = self.mil._mil_desc(cfg["input_dim"]), self.mil._mil_weights()
            need = C.c_size_t()
            _lib.check(self.lib.rrt_mil_workspace_size(C.byref(self.mdesc), self.n, C.byref(need)), "mil workspace")
            self.wss = [torch.empty(need.value, dtype=torch.uint8, device=dev) for _ in range(S)]
            self.outs = [torch.empty(2, dtype=torch.float32, device=dev) for _ in range(S)]
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
        self.enc._desc.compute = self.compute
        self.w = self.enc._weights()
        # one event pair per launch of the dominant kernel in the timed region: every step, every stream
        # (an event pair costs the stream two marker packets: measured ~5 us per forward, 2 % of an fp32 bag and 10 % of a
        #  bf16 one -- so every EV_EVERY-th step is instrumented, at least 8 steps)
        self.ev_every = max(1, min(4, args.steps // 8))
        self.ev_pairs = [(self.hev.create(), self.hev.create()) for _ in range(((args.steps + self.ev_every - 1) // self.ev_every) * S)]
        self.ev_arr = (C.c_void_p * _lib.EV_COUNT)()
        # optional phase gate (RRT_BENCH_GATE=1): the bags' MFMA-bound R-MSA cores take turns instead of time-slicing.
        # Off by default since round 2: with the denser kernels free-running streams are faster at every S
        # (fp32 S=2: 4.53 k vs 4.40 k slides/s, bf16: 12.5 k vs 11.7 k -- DESIGN.md section 5)
        self.gate = C.c_void_p()
        want_gate = os.environ.get("RRT_BENCH_GATE") == "1"
        if want_gate and self.mil is None:
            _lib.check(self.lib.rrt_phase_gate_create(C.byref(self.gate)), "phase gate")
        self.extra = {}
        self._w16_mode = [None] * S      # compute mode of the 16-bit weight images in each stream's workspace

    def _mark(self, a, b):
        for j in range(self._lib.EV_COUNT):
            self.ev_arr[j] = None
        self.ev_arr[self._lib.EV_LN_PARTITION], self.ev_arr[self._lib.EV_QKV] = a, b
        return self.ev_arr

    def step(self, i, timed):
        lib, _lib = self.lib, self._lib
        for s_ in range(self.S):
            x = self.bags[(i * self.S + s_) % len(self.bags)]
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
            evs = self._mark(*self.ev_pairs[(i // self.ev_every) * self.S + s_]) if (timed and i % self.ev_every == 0) else None
            # reduced-precision modes: this stream's workspace keeps the 16-bit weight images of the (unchanged)
            # weights from its first call on, as rrt_mil_amd.RRTEncoder does between forwards (weights16_valid)
            mode = self.enc._desc.compute
            self.enc._desc.weights16_valid = int(mode != _lib.COMPUTE_F32 and self._w16_mode[s_] == mode)
            self._w16_mode[s_] = mode
            rc = lib.rrt_encoder_forward_gated_f32(C.byref(self.enc._desc), C.byref(self.w), x.data_ptr(),
                                                   self.outs[s_].data_ptr(), self.n, self.wss[s_].data_ptr(),
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

    def finish(self, args, world, rank, elapsed):
        import numpy as np
        torch = self.torch
        for o in self.outs:
            assert torch.isfinite(o).all()
        if rank != 0:
            return {}
        rec = {}
        flops, g = fused_flops(self.n, self.enc_cfg)
        peak = PEAK_TFLOPS["bf16" if self.dtype in ("bf16", "f16") else "f32"]
        if self.dtype == "f32x3":
            # projection: 3 bf16 MFMAs per product; attention: fp32 MFMA.  The roofline of THIS arithmetic is the sum of
            # the two parts' matrix-pipe times; `peak` = algorithmic FLOPs over that time
            f_proj, f_attn = 2.0 * g.Np * 1536 * DIM, 4.0 * g.Np * g.P * DIM
            peak = round(flops / (3.0 * f_proj / PEAK_TFLOPS["bf16"] + f_attn / PEAK_TFLOPS["f32"]), 1)
        iso_ms = self.isolated_fused_ms()
        kernel = (f"rmsa_fused_kernel (R-MSA per (region, head): qkv projection {g.P}x192x512 + EPEG + softmax(QK^T)V "
                  f"from LDS; {2.0 * g.Np * 1536 * DIM / 1e9:.2f} + {4.0 * g.Np * g.P * DIM / 1e9:.2f} GFLOP), {self.dtype} operands"
                  + (" (projection: fp32 emulated by 3 bf16 MFMAs per product; attention: fp32 MFMA; peak = FLOPs over "
                     "3 x projection / 2.5 PFLOP/s + attention / 157.3 TFLOP/s)" if self.dtype == "f32x3" else ""))
        if self.mil is None:
            ms = float(np.mean([self.hev.elapsed_ms(a, b) for a, b in self.ev_pairs]))
            ach = flops / (ms * 1e-3) / 1e12
            rec["roofline"] = {"bound": "mfma", "kernel": kernel, "achieved": round(ach, 2), "peak": peak,
                               "unit": "TFLOP/s", "frac": round(ach / peak, 4), "flops_per_launch": flops,
                               "avg_launch_ms": round(ms, 5),
                               "traffic": TRAFFIC_BYTES_PER_LAUNCH.get((self.dtype, self.n)),
                               "note": f"mean over {len(self.ev_pairs)} launches of the timed region (every "
                                       f"{self.ev_every}. step, all streams; {self.S} bag(s) "
                                       "in flight per GPU: the other bag's kernels are co-resident on this launch's "
                                       "CUs for its whole duration and take issue slots from it -- the kernel's own "
                                       "number is roofline_isolated)"}
        ach = flops / (iso_ms * 1e-3) / 1e12
        iso = {"bound": "mfma", "kernel": kernel, "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
               "frac": round(ach / peak, 4), "flops_per_launch": flops, "avg_launch_ms": round(iso_ms, 5),
               "traffic": TRAFFIC_BYTES_PER_LAUNCH.get((self.dtype, self.n)),
               "note": "same kernel, untimed pass with one bag in flight (forwards back to back on one stream after 150 "
                       "forwards of lead, every fourth one instrumented, median of 10 launches)"}
        if self.mil is None:
            rec["roofline_isolated"] = iso
        else:
            rec["roofline"] = iso
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
        loads = [sum(sharding.bag_cost(self.sizes[i], region_num=e["region_num"], epeg_k=e["epeg_k"],
                                       crmsa_k=e["crmsa_k"]) for i in r) for r in self.assign]
        self.extra = {"bags_per_rank": [len(r) for r in self.assign],
                      "cost_imbalance": round(max(loads) / (sum(loads) / len(loads)), 4),
                      "tokens_per_step": int(sum(self.sizes))}

    def step(self, i, timed):
        if self.bags:
            with self.torch.no_grad():
                self.enc.forward_bags(self.bags, streams=self.S, outs=self.outs)

    def sync(self):
        self.torch.cuda.synchronize()

    def finish(self, args, world, rank, elapsed):
        for o in self.outs:
            assert self.torch.isfinite(o).all()
        return {}


# ------------------------------------------------------------------------------------ the rank program
def stabilise(wl, steps_hint, max_s=3.0):
    """Untimed: run probes of a few steps until three consecutive probe rates agree within 3 % (clock ramp,
    first-touch of the workspaces, HIP's launch pipeline) -- so that a short timed region (--steps 20) gives
    the same rate as a long one.  Bounded by max_s seconds.  Returns the number of steps spent."""
    probe = max(4, min(20, steps_hint))
    rates, spent = [], 0
    t_end = time.perf_counter() + max_s
    while time.perf_counter() < t_end:
        wl.sync()
        t0 = time.perf_counter()
        for i in range(probe):
            wl.step(i, False)
        wl.sync()
        rates.append(probe / (time.perf_counter() - t0))
        spent += probe
        if len(rates) >= 3 and max(rates[-3:]) / min(rates[-3:]) < 1.03:
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
    ap.add_argument("--streams", type=int, default=int(os.environ.get("RRT_BENCH_STREAMS", "2")),
                    help="bags in flight per GPU (one HIP stream + workspace each); a step = this many bags")
    ap.add_argument("--stub-cpu", action="store_true", help="rank logic only: CPU stand-in workload over gloo (tests)")
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

    rec_extra = wl.finish(args, world, rank, elapsed)

    if rank == 0:
        dtype = "f32" if args.stub_cpu else wl.dtype
        ms_per_step = elapsed / args.steps * 1e3
        value = wl.units_global * args.steps / elapsed
        enc_cfg = cfg["enc"]
        config = {"workload": cfg["label"] + (" [--stub-cpu: CPU stand-in, rank logic only]" if args.stub_cpu else ""),
                  "baseline_config_index": args.config, "dim": DIM, "bags_per_step": wl.units_global,
                  "streams_per_gpu": getattr(wl, "S", None), "untimed_ramp_steps": ramp,
                  "parallelism": f"bag-parallel x{world} (no data-path collective)"}
        if cfg["n"]:
            config["n_tokens"] = cfg["n"]
            extra_f = 2.0 * cfg["n"] * cfg.get("input_dim", 0) * DIM
            gf = flops_total(cfg["n"], enc_cfg) + extra_f
            config["gflop_per_bag"] = round(gf / 1e9, 2)
            config["whole_path_tflops"] = round(wl.units_global / world * gf / (ms_per_step * 1e-3) / 1e12, 2)
        else:
            gf = sum(flops_total(n, enc_cfg) for n in mix_sizes(cfg["n_bags"]))
            config["gflop_per_step"] = round(gf / 1e9, 2)
            config["whole_path_tflops"] = round(gf / (ms_per_step * 1e-3) / 1e12 / world, 2)
        config.update(wl.extra)
        metric = ("slides/sec RRTEncoder fwd, N=9000 D=512 region_num=8" if args.config == 1
                  else f"slides/sec, BASELINE configs[{args.config}]")
        rec = {"metric": metric, "value": round(value, 2), "unit": "slides/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
               "scaling": wl.scaling, "vs_baseline": None, "dtype": dtype,
               "data": "stub" if args.stub_cpu else "synthetic", "config": config}
        rec.update(rec_extra)
        if not args.stub_cpu and args.config == 1 and world == 1 and not args.no_extras:
            rec.update(extras(wl, dev))
        if not args.stub_cpu and world == 1 and not args.no_cpu_baseline:
            n_cpu = cfg["n"] or 9000          # config 4: the typical bag of the mix
            rec["cpu_baseline"] = cpu_baseline(n_cpu, enc_cfg)
        print(json.dumps(rec), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def extras(wl, dev):
    """Informational records of the default run (rank 0, after the timed region; none is the headline value)."""
    import torch
    from rrt_mil_amd import RRTEncoder, RRTMIL, _lib, synth
    out = {}
    S = wl.S
    # the same workload with bf16 arithmetic (the reference's --amp / autocast path, BASELINE configs[2..4])
    wl.enc._desc.compute = _lib.COMPUTE_BF16
    for i in range(10):
        wl.step(i, False)
    torch.cuda.synchronize()
    ta = time.perf_counter()
    for i in range(40):
        wl.step(i, False)
    torch.cuda.synchronize()
    out["amp_bf16"] = {"value": round(S * 40 / (time.perf_counter() - ta), 2), "unit": "slides/s", "n_gpus": 1,
                       "note": "rank 0 only, 40 steps after the timed region; RRT_COMPUTE_BF16 (bf16 operands on the "
                               "matrix cores, fp32 accumulate; `--config 3` / `--dtype bf16` give the full record)"}
    # fp32 EMULATED on the bf16 matrix cores for the two big projections (RRT_COMPUTE_F32X3; ~1e-6 from the exact path)
    y_exact = wl.outs[0].clone()
    wl.enc._desc.compute = _lib.COMPUTE_F32X3
    for i in range(10):
        wl.step(i, False)
    torch.cuda.synchronize()
    tb = time.perf_counter()
    for i in range(40):
        wl.step(i, False)
    torch.cuda.synchronize()
    tb = time.perf_counter() - tb
    wl.enc._desc.compute = _lib.COMPUTE_F32
    wl.step(39, False)                        # the same bags through the exact path: outs[0] of both modes side by side
    torch.cuda.synchronize()
    y_x3 = y_exact                            # (outs[0] after the timed region held the exact result of another bag)
    wl.enc._desc.compute = _lib.COMPUTE_F32X3
    wl.step(39, False)
    torch.cuda.synchronize()
    y_x3 = wl.outs[0].clone()
    wl.enc._desc.compute = _lib.COMPUTE_F32
    wl.step(39, False)
    torch.cuda.synchronize()
    out["f32x3"] = {"value": round(S * 40 / tb, 2), "unit": "slides/s", "n_gpus": 1,
                    "max_abs_vs_exact_f32": float((y_x3 - wl.outs[0]).abs().max()),
                    "note": "rank 0 only, 40 steps after the timed region; RRT_COMPUTE_F32X3: qkv / proj GEMMs of the R-MSA "
                            "layers as three bf16 MFMAs per product on (hi, lo) bf16 operand pairs, fp32 accumulate; attention, "
                            "LayerNorm, CR-MSA exact fp32 (`--dtype f32x3` gives the full record)"}

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
        (tenc(xg) * gy).sum().backward()
    for _ in range(3):
        tstep()
    torch.cuda.synchronize()
    tt = time.perf_counter()
    for _ in range(20):
        tstep()
    torch.cuda.synchronize()
    tt = (time.perf_counter() - tt) / 20
    out["train_step"] = {"ms_per_step": round(tt * 1e3, 4), "steps_per_s": round(1.0 / tt, 2),
                         "peak_mem_mb": round(torch.cuda.max_memory_allocated(dev) / 1e6, 1),
                         "note": "RRTEncoder.train() forward (stash) + backward of every parameter, N=9000 D=512, fp32, "
                                 "drop_out=0.1, one bag per step (rrt_encoder_forward_train_f32 / rrt_encoder_backward_f32)"}
    return out


if __name__ == "__main__":
    main()
