#!/usr/bin/env python3
"""bench.py -- slides/sec of the RRTEncoder forward on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one batch of `--streams` (default 2) independent bags per GPU, each one
RRTEncoder forward (eval, fp32) over a device-resident synthetic bag of N=9000 x D=512
(BASELINE.json configs[1], the config the metric is quoted on), each on its own HIP stream
with its own workspace: bags are independent units, so a second bag's
memory-bound and small kernels run in the gaps of the first bag's MFMA-bound ones.
Bag-parallel across GPUs too: every rank owns its own bags, no data-path collective
(scaling = "weak"); the only collectives are the barrier and a MAX of the elapsed time.

Besides the contract fields the JSON line carries
  roofline     -- the dominant kernel (the fused R-MSA kernel, fp32 MFMA): algorithmic FLOPs
                  per launch / its average duration, measured live with HIP events that
                  librrt_hip records on the launch stream inside the timed region;
  cpu_baseline -- the oracle's eager torch-CPU port of the reference op sequence
                  (oracle/rrt_oracle.py::forward_eager) timed on this box's host cores
                  over a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) in creation order.  With the
# default, the streams RCCL creates under torch.distributed.run shift that mapping so that the bags' two
# streams land on ONE hardware queue and serialise (measured: 3.72 k slides/s instead of 4.33 k, exactly the
# one-stream rate).  8 queues keeps every stream of this process on its own queue in both launch modes.
# Must be set before the HIP runtime initialises, i.e. before `import torch`.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from rrt_mil_amd import RRTEncoder, _lib, sharding, synth  # noqa: E402
from rrt_mil_amd.geometry import region_grid  # noqa: E402

N_TOKENS, DIM = 9000, 512
CFG = dict(mlp_dim=512, epeg_k=15, crmsa_k=3, region_num=8)
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, 2.4 GHz
# HBM-side bytes of the dominant kernel (rmsa_fused_kernel) per launch from rocprofv3 --pmc (separate
# FETCH_SIZE and WRITE_SIZE passes; FETCH_SIZE doubled per the gfx950 correction):
# 2 * 33965.7 KiB + 18432 KiB -- profiles/r01_e_fused_traffic_pmc.txt
TRAFFIC_BYTES_PER_LAUNCH = 88.4e6


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


def flops_total(n):
    """Algorithmic FLOPs per bag, SURVEY.md §8(d)."""
    g, g8 = region_grid(n, CFG["region_num"]), region_grid(n, 8)
    k, h, ek, D = CFG["crmsa_k"], 8, CFG["epeg_k"], DIM
    return (8 * g.Np * D * D + 4 * g.Np * g.P * D + 2 * g.Np * g.P * h * ek
            + 6 * g8.Np * D * k + 8 * k * 64 * D * D + 4 * k * 64 * 64 * D)


def cpu_baseline(budget_s=22.0):
    """Reference-equivalent CPU path (oracle port) on this host: bounded sample.  The eager op
    sequence scales poorly past a few dozen threads (many small aten ops), so a short probe picks
    the thread count at which the reference path is FASTEST before the timed sample."""
    from oracle import rrt_oracle  # the only place bench.py touches oracle/
    state = synth.encoder_state(**CFG)
    st = {k: torch.from_numpy(v) for k, v in state.items()}
    x = torch.from_numpy(synth.bag(N_TOKENS, DIM))
    ncpu = os.cpu_count() or torch.get_num_threads()
    cands = sorted({c for c in (ncpu, ncpu // 2, ncpu // 4, 32, 16, 8) if 1 <= c <= ncpu}, reverse=True)
    probe = {}
    for c in cands:
        torch.set_num_threads(c)
        rrt_oracle.forward_eager(x, st, CFG)         # warm-up at this thread count
        t0 = time.perf_counter()
        rrt_oracle.forward_eager(x, st, CFG)
        probe[c] = time.perf_counter() - t0
    cores = min(probe, key=probe.get)
    torch.set_num_threads(cores)
    times = []
    t_end = time.perf_counter() + budget_s * 0.6
    while len(times) < 5 or (time.perf_counter() < t_end and len(times) < 200):
        t0 = time.perf_counter()
        rrt_oracle.forward_eager(x, st, CFG)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": round(1.0 / med, 3), "unit": "slides/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} bags of N={N_TOKENS} D={DIM} (median {med * 1e3:.1f} ms/bag), "
                      f"oracle/rrt_oracle.py::forward_eager (same aten op sequence as the reference, "
                      f"bit-identical to it in the build container), torch {torch.__version__} CPU, "
                      f"{cores} of {ncpu} hardware threads (fastest of "
                      + ", ".join(f"{c}: {probe[c] * 1e3:.0f} ms" for c in cands) + ")"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("RRT_BENCH_STREAMS", "2")),
                    help="bags in flight per GPU (one HIP stream + workspace each); a step = this many bags")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks "
                         f"(WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "TORCHELASTIC_RUN_ID" in os.environ:   # under torch.distributed.run, also at 1 rank
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)   # RCCL

    state = synth.encoder_state(**CFG)
    enc = RRTEncoder(**CFG).eval()
    enc.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()}, strict=True)
    enc = enc.to(dev)
    # a few distinct device-resident bags per rank, cycled (one bag per step)
    bags = [torch.from_numpy(synth.bag(N_TOKENS, DIM, tag=f"bench/r{rank}/b{i}")).to(dev) for i in range(4)]
    out = torch.empty_like(bags[0])

    lib = _lib.load()
    hev = HipEvents()
    S = max(1, args.streams)
    tstreams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(dev) for _ in range(S - 1)]
    streams = [t.cuda_stream for t in tstreams]
    need = enc._workspace(N_TOKENS, dev).numel()
    wss = [torch.empty(need, dtype=torch.uint8, device=dev) for _ in range(S)]   # one workspace per bag in flight
    outs = [torch.empty_like(bags[0]) for _ in range(S)]
    out = outs[0]
    w = enc._weights()
    ev_pairs = [(hev.create(), hev.create()) for _ in range(args.steps)]
    ev_arr = (C.c_void_p * _lib.EV_COUNT)()

    # bags in flight on different streams share a phase gate: their MFMA-bound R-MSA cores take turns
    # instead of time-slicing the matrix pipes, and the other bag's memory-bound kernels fill the gaps
    gate = C.c_void_p()
    if (S == 2 or os.environ.get("RRT_BENCH_GATE") == "1") and os.environ.get("RRT_BENCH_GATE", "1") != "0":
        # (the gate pays at two bags in flight; with three or more, free-running streams are faster -- DESIGN.md §5)
        _lib.check(lib.rrt_phase_gate_create(C.byref(gate)), "phase gate")

    def step(i, timed):
        # one step = S independent bags, one per stream (bag-parallel inside the GPU as well)
        for s_ in range(S):
            x = bags[(i * S + s_) % len(bags)]
            evs = None
            if timed and s_ == 0:   # mark the dominant kernel: [after LN+partition, after the fused R-MSA core]
                for j in range(_lib.EV_COUNT):
                    ev_arr[j] = None
                ev_arr[_lib.EV_LN_PARTITION] = ev_pairs[i][0]
                ev_arr[_lib.EV_QKV] = ev_pairs[i][1]
                evs = ev_arr
            rc = lib.rrt_encoder_forward_gated_f32(C.byref(enc._desc), C.byref(w), x.data_ptr(),
                                                   outs[s_].data_ptr(), N_TOKENS, wss[s_].data_ptr(),
                                                   wss[s_].numel(), streams[s_], gate, evs)
            _lib.check(rc, "forward")

    for i in range(args.warmup):
        step(i, False)
    torch.cuda.synchronize()
    # untimed reference pass: the dominant kernel alone on the chip (one bag in flight), so that
    # its roofline fraction can also be read without the co-running bags of the timed region
    iso_pairs = [(hev.create(), hev.create()) for _ in range(10)]
    for a, b in iso_pairs:
        for j in range(_lib.EV_COUNT):
            ev_arr[j] = None
        ev_arr[_lib.EV_LN_PARTITION], ev_arr[_lib.EV_QKV] = a, b
        _lib.check(lib.rrt_encoder_forward_events_f32(C.byref(enc._desc), C.byref(w), bags[0].data_ptr(),
                                                      outs[0].data_ptr(), N_TOKENS, wss[0].data_ptr(),
                                                      wss[0].numel(), streams[0], ev_arr), "forward")
        torch.cuda.synchronize()
    iso_ms = float(np.median([hev.elapsed_ms(a, b) for a, b in iso_pairs]))
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed = sharding.max_over_ranks(elapsed, device=dev)   # whole-job time = slowest rank
    assert torch.isfinite(out).all()

    # informational: the same workload with bf16 Linear operands (the reference's --amp / autocast
    # path, BASELINE configs[2..4]); fp32 accumulate and fp32 tensors in HBM.  Not the headline value.
    amp = None
    if rank == 0:
        enc._desc.compute = _lib.COMPUTE_BF16
        for i in range(5):
            step(i, False)
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for i in range(30):
            step(i, False)
        torch.cuda.synchronize()
        amp = {"value": round(S * 30 / (time.perf_counter() - ta), 2), "unit": "slides/s", "n_gpus": 1,
               "note": "rank 0 only, 30 steps after the timed region; RRT_COMPUTE_BF16 (bf16 MFMA operands in "
                       "the Linear layers / fused projection, fp32 accumulate)"}
        enc._desc.compute = _lib.COMPUTE_F32

    # informational: the whole slide classifier of BASELINE configs[2] (C16-R50 shape) through the one-call
    # path (row f1): N=9000 x 1024 features -> fc 512 + ReLU -> encoder(crmsa_k=1, all_shortcut) ->
    # DAttention -> predictor; one bag in flight, fp32.  Not the headline value.
    mil_rec = None
    if rank == 0:
        from rrt_mil_amd import RRTMIL
        mcfg = dict(input_dim=1024, n_classes=2, epeg_k=15, crmsa_k=1, all_shortcut=True)
        mil = RRTMIL(**mcfg).eval()
        mst = synth.mil_state(input_dim=1024, n_classes=2, epeg_k=15, crmsa_k=1)
        mil.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in mst.items()}, strict=True)
        mil = mil.to(dev)
        feats = torch.from_numpy(synth.bag(N_TOKENS, 1024, tag="mil", nonneg=True)).to(dev).unsqueeze(0)
        for _ in range(5):
            lg = mil(feats)
        torch.cuda.synchronize()
        tm = time.perf_counter()
        for _ in range(50):
            lg = mil(feats)
        torch.cuda.synchronize()
        tm = (time.perf_counter() - tm) / 50
        assert torch.isfinite(lg).all()
        mil_rec = {"value": round(1.0 / tm, 2), "unit": "slides/s", "ms_per_slide": round(tm * 1e3, 4), "n_gpus": 1,
                   "note": "RRTMIL(input_dim=1024, epeg_k=15, crmsa_k=1, all_shortcut=True).eval() forward, "
                           "N=9000 x 1024 -> logits, fp32, one bag in flight, rank 0 after the timed region "
                           "(rrt_mil_forward_f32: patch_to_emb GEMM+ReLU, encoder, DAttention pooling, predictor)"}
        del mil, feats

    # informational: one training step of the same encoder (row f2): forward with stash + full backward, fp32,
    # default proj dropout 0.1, one bag per step, rank 0 after the timed region.  Not the headline value.
    train_rec = None
    if rank == 0:
        tenc = RRTEncoder(**CFG).to(dev).train()
        tenc.load_state_dict(enc.state_dict())
        xg = bags[0].unsqueeze(0)
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
        train_rec = {"ms_per_step": round(tt * 1e3, 4), "steps_per_s": round(1.0 / tt, 2),
                     "peak_mem_mb": round(torch.cuda.max_memory_allocated(dev) / 1e6, 1),
                     "note": "RRTEncoder.train() forward (stash) + backward of every parameter, N=9000 D=512, fp32, "
                             "drop_out=0.1, one bag per step (rrt_encoder_forward_train_f32 / rrt_encoder_backward_f32)"}
        del tenc

    # dominant kernel: rmsa_fused_kernel = qkv projection [Np, D] x [3D, D]^T + region attention
    # (Q K^T and A V) per (region, head), fp32 MFMA.  Algorithmic FLOPs per launch (SURVEY §8d terms):
    g = region_grid(N_TOKENS, CFG["region_num"])
    qkv_flops = 2.0 * g.Np * (3 * DIM) * DIM + 4.0 * g.Np * g.P * DIM
    qkv_ms = float(np.mean([hev.elapsed_ms(a, b) for a, b in ev_pairs]))
    achieved = qkv_flops / (qkv_ms * 1e-3) / 1e12

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * S * args.steps / elapsed
        rec = {
            "metric": "slides/sec RRTEncoder fwd, N=9000 D=512 region_num=8",
            "value": round(value, 2), "unit": "slides/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: RRTEncoder(mlp_dim=512, epeg_k=15, crmsa_k=3, "
                                   "region_num=8).eval() forward, one device-resident bag N=9000 D=512 "
                                   "per GPU per step, fp32, closed-form weights",
                       "n_tokens": N_TOKENS, "dim": DIM, "bags_per_step": world * S, "streams_per_gpu": S,
                       "parallelism": f"bag-parallel x{world} (no data-path collective)",
                       "gflop_per_bag": round(flops_total(N_TOKENS) / 1e9, 2),
                       "whole_path_tflops": round(S * flops_total(N_TOKENS) / (ms_per_step * 1e-3) / 1e12, 2)},
            "roofline": {"bound": "mfma", "kernel": "rmsa_fused_kernel<9,0> (R-MSA per (region, head): qkv projection 144x192x512 + EPEG + "
                                   "softmax(QK^T)V from LDS; 14.50 + 2.72 GFLOP)",
                         "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         "flops_per_launch": qkv_flops, "avg_launch_ms": round(qkv_ms, 5),
                         "traffic": TRAFFIC_BYTES_PER_LAUNCH,
                         "note": f"measured over the timed region with {S} bag(s) in flight per GPU: the launch "
                                 "shares the chip with the other bag's kernels (see roofline_isolated)"},
            "roofline_isolated": {"bound": "mfma", "achieved": round(qkv_flops / (iso_ms * 1e-3) / 1e12, 2),
                                  "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                  "frac": round(qkv_flops / (iso_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                  "avg_launch_ms": round(iso_ms, 5),
                                  "note": "same kernel, untimed pass with one bag in flight (median of 10)"},
        }
        rec["amp_bf16"] = amp
        rec["rrtmil_c16"] = mil_rec
        rec["train_step"] = train_rec
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline()
        print(json.dumps(rec), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
