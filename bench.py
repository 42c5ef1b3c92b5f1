#!/usr/bin/env python3
"""bench.py -- VSOM training hot path on MI355X.

Metric (BASELINE.json): training samples/sec per epoch (BMU + update), 128x128 map, 784-dim.
A "step" = one Som::trainBatchSomEpoch (Som.cpp:756-879) pass -- full BMU search (is_first) +
neighbourhood mean/sigma^2 update -- over one chunk of B=4096 synthetic MNIST-like samples per
GPU, chunk already resident in HBM when the timed region starts.  With N GPUs the chunk is
4096*N samples (weak scaling): phase 1 shards samples, phase 2 shards nodes, RCCL all-gathers
exchange lastBMU / the new map rows (variational-self-organizing-maps_amd/dist.py).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant
kernel = the phase-2 update kernel, fp32 VALU work priced against the 157.3 TFLOP/s fp32 peak
that gfx950's vector and matrix pipes share) and `cpu_baseline` (the CPU oracle timed on this
box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: Peak FP32 vector == matrix (dense)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--map", type=int, default=128, help="map side (default 128 -> 128x128)")
    ap.add_argument("--dim", type=int, default=784)
    ap.add_argument("--chunk", type=int, default=4096, help="samples per GPU per step")
    ap.add_argument("--sigma", type=float, default=32.0)
    ap.add_argument("--strong", action="store_true", help="fixed total chunk (strong scaling)")
    ap.add_argument("--local", action="store_true", help="time the later-epoch (findLocalBmu) pass")
    ap.add_argument("--fma", action="store_true",
                    help="opt-in contracted update arithmetic (within 1e-5 of the reference, not bit-identical)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--nchunks", type=int, default=4, help="distinct resident chunks cycled over")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N>1 (gloo only to rehearse the N>1 flow on one GPU)")
    ap.add_argument("--share-device", action="store_true",
                    help="rehearsal: every rank uses cuda:0 (RCCL refuses that, so combine with --backend gloo)")
    ap.add_argument("--host-chunks", choices=["off", "sync", "overlap"], default="off",
                    help="PCIe-inclusive variants (N=1, not the contract number): chunks start in pinned host "
                         "memory every step; 'sync' = vsom_upload_chunk, 'overlap' = prefetch of chunk i+1 "
                         "beside the epoch of chunk i (vsom_prefetch_chunk / vsom_commit_chunk)")
    return ap.parse_args()


def cpu_baseline(args, X_host, init_map):
    """Oracle (CPU port of the reference algorithm) on a bounded prefix of one chunk."""
    from oracle import pyoracle as po
    W = H = args.map
    cores = po.max_threads()
    B = X_host.shape[0]

    def run(nsamp, threads, faithful=False):
        o = po.OracleSom(W, H, args.dim, po.STANDARD)
        o.set_state(map=init_map)
        lb = np.zeros(nsamp, np.uint64)
        t0 = time.perf_counter()
        o.batch_epoch(X_host[:nsamp], lb, args.sigma, True, nthreads=threads, faithful=faithful)
        dt = time.perf_counter() - t0
        o.close()
        return dt

    # calibrate on a small prefix, then size the sample for ~cpu_seconds of CPU work
    n0 = min(B, 8 * cores)
    t_small = run(n0, cores)
    n = int(min(B, max(n0, n0 * args.cpu_seconds * 0.6 / max(t_small, 1e-3))))
    t_lean = run(n, cores)
    lean = n / t_lean
    # the reference as shipped is serial and allocates per call: time that shape on a few samples
    nf = max(2, min(n, int(n0 // 4)))
    t_f = run(nf, 1, faithful=True)
    faithful = nf / t_f
    best = max(lean, faithful)
    return {
        "value": round(best, 3), "unit": "samples/s",
        "cores": cores if lean >= faithful else 1, "kind": "port",
        "sample": (f"oracle vso_batch_epoch (findBmu + update) on the first {n} samples of one "
                   f"{B}-sample chunk, {W}x{H}x{args.dim} map, OpenMP over samples/nodes"),
        "lean_all_cores_samples_per_s": round(lean, 3),
        "faithful_1thread_samples_per_s": round(faithful, 3),
        "faithful_sample": nf,
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
        args.gpus = world

    import torch
    import torch.distributed as dist
    import gen
    import vsom_amd
    from vsom_amd import capi
    import importlib
    vdist = importlib.import_module("variational-self-organizing-maps_amd.dist")

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    W = H = args.map
    D = args.dim
    Bper = args.chunk
    Bglob = Bper if args.strong else Bper * world

    # synthetic data: MNIST-like uint8-valued floats; distinct chunks, same on all ranks
    chunks_host = [gen.mnist_like(Bglob, seed=3 + i, dim=D) for i in range(args.nchunks)]
    init_map = gen.random_map(W * H, D, seed=42, scale=1.0) * np.float32(100.0) + np.float32(100.0)

    stream = torch.cuda.Stream(device=dev)
    ctx = vsom_amd.Context(W, H, D, capi.STANDARD, device=local_rank)
    ctx.set_state(map=init_map)
    ctx.set_stream(stream.cuda_stream)
    if args.fma:
        ctx.set_update_mode(capi.UPDATE_FMA)
    chunks = [torch.from_numpy(c).to(dev) for c in chunks_host]
    torch.cuda.synchronize()

    eng = vdist.HipEngine(ctx, dev)
    trainer = vdist.ShardedBatchTrainer(eng, rank, world)
    is_first = not args.local

    pinned = None
    if args.host_chunks != "off":
        if world != 1:
            raise SystemExit("--host-chunks is a single-GPU measurement")
        pinned = [capi.PinnedBuffer(c.shape) for c in chunks_host]
        for pb, c in zip(pinned, chunks_host):
            pb.array[...] = c
        if args.host_chunks == "overlap":
            ctx.prefetch_chunk(pinned[0].array)

    def step(i):
        with torch.cuda.stream(stream):
            if args.host_chunks == "sync":
                ctx.upload_chunk(pinned[i % len(pinned)].array)      # blocking H2D + staging
                eng._bind_chunk()
            elif args.host_chunks == "overlap":
                ctx.commit_chunk()                                   # chunk i (copied during step i-1)
                eng._bind_chunk()
            else:
                eng.load_chunk_device(chunks[i % len(chunks)])   # staging + lastBMU reset (DataSet.cpp:118-160)
            trainer.epoch(args.sigma, is_first)
            if args.host_chunks == "overlap":
                ctx.prefetch_chunk(pinned[(i + 1) % len(pinned)].array)   # H2D of chunk i+1 beside this epoch

    for i in range(args.warmup):
        step(i)
    with torch.cuda.stream(stream):
        trainer.flush()
    torch.cuda.synchronize()
    ctx.get_timing(reset=True)
    ctx.enable_timing(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    with torch.cuda.stream(stream):
        trainer.flush()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timing = ctx.get_timing(reset=True)
    ctx.enable_timing(False)

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # secondary, informational: the same step with the opt-in contracted update arithmetic
    # (M = fma(c,d,M), S = fma(w*d,d,S); results within 1e-5 of the reference instead of bit-identical).
    # `value` above is the strict run; this is reported beside it, never instead of it.
    fma_extra = None
    if not args.fma:
        ctx.set_update_mode(capi.UPDATE_FMA)
        for i in range(2):
            step(i)
        with torch.cuda.stream(stream):
            trainer.flush()
        torch.cuda.synchronize()
        ctx.get_timing(reset=True)
        ctx.enable_timing(True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(args.steps):
            step(2 + i)
        with torch.cuda.stream(stream):
            trainer.flush()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dtf = time.perf_counter() - t1
        tf = ctx.get_timing(reset=True)
        ctx.enable_timing(False)
        ctx.set_update_mode(capi.UPDATE_STRICT)
        if world > 1:
            t = torch.tensor([dtf], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dtf = float(t.item())
        fma_extra = (dtf, tf["update"][0] / max(tf["update"][1], 1))

    mse = float(ctx.get_mse())
    sl_stats = ctx.shortlist_stats()
    if rank == 0:
        nloc = W * H // world if (W * H) % world == 0 else None
        n_nodes_rank = (vdist.shard_bounds(W * H, world, 0)[1])
        upd_ms, upd_cnt = timing["update"]
        upd_avg_s = upd_ms / max(upd_cnt, 1) / 1e3
        # algorithmic work of one update launch: 6 flop per (node, dim, sample)  
        flops_launch = 6.0 * n_nodes_rank * D * Bglob
        achieved = flops_launch / upd_avg_s / 1e12 if upd_avg_s > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("update_kernel_hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "training samples/sec per epoch (BMU+update), 128x128 map, 784-dim",
            "value": round(args.steps * Bglob / dt, 3),
            "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if args.host_chunks == "off" else f"synthetic, chunks in pinned host memory ({args.host_chunks})",
            "update_arithmetic": "fma (opt-in, 1e-5 relative)" if args.fma else "strict (bit-identical to the CPU oracle)",
            "config": {"workload": (f"{W}x{H} map, {D}-dim MNIST-like synthetic, standard transformation, "
                                    f"trainBatchSomEpoch({'findBmu' if is_first else 'findLocalBmu'} + update), "
                                    f"chunk B={Bper}/GPU ({Bglob} total), sigma={args.sigma}"),
                       "map": [W, H], "dim": D, "chunk_per_gpu": Bper, "chunk_total": Bglob,
                       "parallelism": "1 GPU" if world == 1 else f"phase1 sample-sharded x{world}, phase2 node-sharded x{world}, RCCL all-gather"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": FP32_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / FP32_PEAK_TFLOPS, 4),
                         "traffic": traffic,
                         "kernel": "vsom_update_std_rd14_gfx950 (phase-2 mean/sigma^2 chains, hand-scheduled)",
                         "note": ("fp32 VALU-bound kernel priced against the fp32 dense peak shared by the vector and "
                                  "matrix pipes; strict non-FMA arithmetic caps it at 0.5"),
                         "avg_launch_ms": round(upd_avg_s * 1e3, 4),
                         "algorithmic_flop_per_launch": flops_launch},
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in timing.items()},
            "mse_last": mse,
            "bmu_shortlist_last": sl_stats,
        }
        if fma_extra is not None:
            dtf, upd_f_ms = fma_extra
            ach_f = flops_launch / (upd_f_ms / 1e3) / 1e12 if upd_f_ms > 0 else 0.0
            out["fma_mode"] = {"note": "opt-in VSOM_UPDATE_FMA arithmetic (tests/test_gpu_fma_mode.py: map/sigma within "
                                       "1e-5 relative -- measured 3.4e-7 at this size, tools/fma_error_report.py -- BMU "
                                       "indices / bmuHits / MSE / weightMap bit-exact); not the headline",
                               "value": round(args.steps * Bglob / dtf, 3), "ms_per_step": round(dtf / args.steps * 1e3, 4),
                               "update_avg_launch_ms": round(upd_f_ms, 4), "update_achieved_tflops": round(ach_f, 3),
                               "update_frac_of_peak": round(ach_f / FP32_PEAK_TFLOPS, 4)}
        if not args.no_cpu and world == 1:   # the CPU leg runs at N=1 only (contract)
            out["cpu_baseline"] = cpu_baseline(args, chunks_host[0][:Bper], init_map)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
