#!/usr/bin/env python3
"""bench.py -- VSOM training hot path on MI355X.

Metric (BASELINE.json): training samples/sec per epoch (BMU + update), 128x128 map, 784-dim.
A "step" = one Som::trainBatchSomEpoch (Som.cpp:756-879) pass -- full BMU search (is_first) +
neighbourhood mean/sigma^2 update -- over one chunk of B=4096 synthetic MNIST-like samples per
GPU, the rank's own samples already resident in HBM when the timed region starts.

  python bench.py --gpus 1 --steps 20 --warmup 3            (default: C3, the BASELINE headline)
  python bench.py --config c2|c4|c5|online                  (the other BASELINE.json configs, 1 GPU)
  python bench.py --gpus N                                  (starts N ranks itself, see below)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N > 1: BASELINE.json's config 3 is "128x128 map, batch=4096 sharded across 8 GPUs", so `value` is
the STRONG split -- the 4096-sample chunk of the step is shared out, rank r OWNS rows [4096 r/N,
4096 (r+1)/N) -- and `scaling` says "strong".  (One chunk of 4096*N samples would be a different
algorithmic step: every chunk OVERWRITES the map with its own weighted mean, Som.cpp:870.)  The weak
variant (every rank owns 4096 samples of a 4096*N chunk) is timed in the same run and reported as
`weak_scaling`; --scaling weak makes it the `value`.  Inside the timed step the chunk is replicated
with an all-gather of X (the node-sharded phase 2 reads every sample), phase 1 runs on the rank's
samples, lastBMU / ||residual||^2 are all-gathered, phase 2 runs on the rank's 16384/N nodes and
the new map rows are all-gathered (variational-self-organizing-maps_amd/dist.py; `backend` names the
transport, `rccl_ranks` counts ranks only when it is RCCL).  When WORLD_SIZE is unset and --gpus
N > 1 this script starts the N ranks itself (python -m torch.distributed.run as a child process,
before anything in this process touches the GPU) and exits with the child's status.
--group times the other multi-GPU front end instead: ONE process, vsom_group_batch_epoch_async over
the N devices (csrc/vsom_group.hip -- what a C++ caller of Som::train gets with VSOM_DEVICES set).

Rank 0 prints ONE JSON line (contract in the task statement) with `roofline` (dominant kernel of
the configuration, HIP-event time on the stream it runs on) and `cpu_baseline` (the CPU oracle
timed on this box's host cores on a bounded sample of the same workload).

Arithmetic of the update chains (--arith, include/vsom_hip.h vsom_update_mode): `strict` (default, and the
library's default: one rounding per fp32 operation, everything bit-identical to the reference's SSE2
arithmetic over whole schedules -- what `value` is quoted on), `sigma` (only S = fma(w*d,d,S) contracted:
map, BMU indices, bmuHits, MSE, weightMap still bit-identical over whole schedules, sigmaMap within 1e-5
relative; tests/test_gpu_fma_schedule.py) or `contracted` (M = fma(c,d,M) too: within 1e-5 for ONE epoch
from a given map, but a schedule leaves the reference's trajectory -- profiles/r3_fma_schedule.jsonl -- so
it is a throughput figure, not a parity mode).  The other two are timed in the same run and reported
beside the headline (`other_arithmetics`).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: peak FP32, vector == matrix (dense)
HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E ~8 TB/s

# BASELINE.json configs.  transform: 0 Standard, 1 Median, 2 CLR (vsom_hip.h)
CONFIGS = {
    "c3": dict(map=128, dim=784, chunk=4096, sigma=32.0, transform=0, data="mnist",
               name="128x128 map, 784-dim MNIST-like synthetic, standard transformation"),
    "c2": dict(map=64, dim=784, chunk=4096, sigma=16.0, transform=0, data="mnist",
               name="64x64 map, 784-dim MNIST-like synthetic, standard transformation"),
    "c4": dict(map=64, dim=32, chunk=16384, sigma=16.0, transform=1, data="blobs",
               name="64x64 map, 32-dim synthetic blobs, median-estimator transformation"),
    "c5": dict(map=32, dim=64, chunk=8192, sigma=8.0, transform=2, data="correlated",
               name="32x32 map, 64-dim correlated synthetic, combinatorial-linear-regression transformation (D=4032)"),
    "online": dict(map=128, dim=784, chunk=512, sigma=8.0, transform=0, data="mnist",
                   name="128x128 map, 784-dim MNIST-like synthetic, standard transformation, online "
                        "trainSingle steps (Exponential decay, eta=0.1)"),
}
METRIC = "training samples/sec per epoch (BMU+update), 128x128 map, 784-dim"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS),
                    help="BASELINE.json configuration (default c3 = the headline, 128x128x784)")
    ap.add_argument("--map", type=int, default=None, help="map side (overrides the config)")
    ap.add_argument("--dim", type=int, default=None, help="sample length J (overrides the config)")
    ap.add_argument("--chunk", type=int, default=None, help="samples per GPU per step (overrides the config)")
    ap.add_argument("--sigma", type=float, default=None)
    ap.add_argument("--scaling", choices=["auto", "strong", "weak"], default="auto",
                    help="N > 1: what `value` is quoted on.  auto = strong (BASELINE config 3: the 4096-sample chunk "
                         "sharded across the GPUs); the other split is timed too and reported beside it")
    ap.add_argument("--strong", action="store_true", help="alias of --scaling strong")
    ap.add_argument("--no-other-scaling", action="store_true", help="skip the second (other-split) timed run at N > 1")
    ap.add_argument("--group", action="store_true",
                    help="N > 1 in ONE process through vsom_group_* (csrc/vsom_group.hip) instead of one process per "
                         "GPU over torch.distributed; with --share-device every member uses device 0 (rehearsal)")
    ap.add_argument("--local", action="store_true", help="time the later-epoch (findLocalBmu) pass")
    ap.add_argument("--arith", choices=["strict", "sigma", "contracted"], default="strict",
                    help="arithmetic of the update chains for `value` (the others are reported beside it)")
    ap.add_argument("--no-other-arith", action="store_true", help="skip the second (other-arithmetic) timed run")
    ap.add_argument("--data", choices=list(DATA_VARIANTS), default="uint8_sparse",
                    help="c2 / c3: what the rows hold for `value` (default: the raw uint8-valued pixels of the reference's "
                         "MNIST loader); the other two are timed in the same run and reported as `data_variants`")
    ap.add_argument("--no-data-variants", action="store_true", help="skip the timed runs on the other data kinds")
    ap.add_argument("--no-stage-ahead", action="store_true",
                    help="N = 1 batch steps: stage every chunk at the start of its own step (rounds 1-4) instead of beside "
                         "the chains of the previous step (vsom_stage_next_device / vsom_commit_chunk)")
    ap.add_argument("--online-search", choices=["auto", "exact", "image"], default="auto",
                    help="--config online: how the chunk loop finds each sample's BMU (vsom_set_bmu_mode): exact = the fp32 "
                         "scan of the whole map per sample; image = the one-byte image + exact refinement; auto = the library's choice")
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
    args = ap.parse_args()
    cfg = dict(CONFIGS[args.config])
    for k in ("map", "dim", "chunk", "sigma"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)
    args.cfg = cfg
    if args.strong:
        args.scaling = "strong"
    if args.scaling == "auto":
        args.scaling = "strong"
    return args


def self_launch(args):
    """--gpus N without a launcher: start the N ranks as a child torch.distributed.run.  Nothing in
    this process has touched the GPU (only numpy is imported), and the child is a new process, not an
    exec of this one."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


DATA_VARIANTS = ("uint8_sparse", "float_sparse", "float_dense")
DATA_TEXT = {
    "uint8_sparse": ("raw MNIST-like pixels 0..255 as MnistDataLoader.cpp:73-75 yields them (the headline): integer "
                     "contraction on one int8 digit per sample value, dead columns retired, zero-quad form"),
    "float_sparse": ("the same pixels normalised to [0,1] (x/255, float-valued): three-digit integer contraction, dead "
                     "columns retired, zero-quad form"),
    "float_dense": ("signed dense float rows (8 blobs, unit sigma 0.5): none of the exact data-dependent shortcuts applies "
                    "-- no dead column, no all-zero quad, no uint8 digit"),
}


def make_data(cfg, n, seed, variant="uint8_sparse"):
    import gen
    if cfg["data"] == "mnist" and variant == "float_dense":
        return gen.float_dense(n, seed=seed + 4, dim=cfg["dim"])
    if cfg["data"] == "mnist" and variant == "float_sparse":
        return (make_data(cfg, n, seed) / np.float32(255.0)).astype(np.float32)
    if cfg["data"] == "mnist":
        d = os.environ.get("VSOM_MNIST_DIR")
        if d:       # the real training images when the IDX files are at hand (none in the build image)
            x = gen.mnist_idx(d, n, offset=(seed - 3) * n, dim=cfg["dim"])
            if x is not None:
                return x
        return gen.mnist_like(n, seed=seed, dim=cfg["dim"])
    if cfg["data"] == "blobs":
        return gen.blobs(n, cfg["dim"], 8, 1, seed, sigma=1.0)
    return gen.correlated(n, cfg["dim"], seed)


def make_map(cfg, depth, variant="uint8_sparse"):
    import gen
    n = cfg["map"] * cfg["map"]
    if cfg["data"] == "mnist" and variant != "float_dense":
        m = gen.random_map(n, depth, seed=42, scale=1.0) * np.float32(100.0) + np.float32(100.0)
        return m if variant == "uint8_sparse" else (m / np.float32(255.0)).astype(np.float32)
    return gen.random_map(n, depth, seed=42)


def cpu_baseline(args, X_host, init_map, online=False):
    """Oracle (CPU port of the reference algorithm) on a bounded prefix of one chunk."""
    from oracle import pyoracle as po
    cfg = args.cfg
    W = H = cfg["map"]
    cores = po.max_threads()
    B = X_host.shape[0]

    def run(nsamp, threads, faithful=False):
        o = po.OracleSom(W, H, cfg["dim"], cfg["transform"])
        o.set_state(map=init_map)
        lb = np.zeros(nsamp, np.uint64)
        t0 = time.perf_counter()
        if online:
            o.train_online_chunk(X_host[:nsamp], lb, 0.1, cfg["sigma"], po.EXPONENTIAL)
        else:
            o.batch_epoch(X_host[:nsamp], lb, cfg["sigma"], True, nthreads=threads, faithful=faithful)
        dt = time.perf_counter() - t0
        o.close()
        return dt

    if online:
        # strictly sequential in samples: one thread (the oracle's trainSingle is scalar code)
        n0 = min(B, 4)
        t_small = run(n0, 1)
        n = int(min(B, max(n0, n0 * args.cpu_seconds * 0.6 / max(t_small, 1e-3))))
        t = run(n, 1)
        return {"value": round(n / t, 3), "unit": "samples/s", "cores": 1, "kind": "port",
                "sample": (f"oracle vso_train_online_chunk (trainSingle x {n}) on the first {n} samples of one "
                           f"{B}-sample chunk, {W}x{H}x{cfg['dim']} map, sigma={cfg['sigma']}")}
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
                   f"{B}-sample chunk, {W}x{H}x{cfg['dim']} map, OpenMP over samples/nodes"),
        "lean_all_cores_samples_per_s": round(lean, 3),
        "faithful_1thread_samples_per_s": round(faithful, 3),
        "faithful_sample": nf,
    }


def profile_traffic(key):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/traffic.json, written by tools/collect_profiles.sh + tools/pmc_summary.py): counters cannot
    be read inside this run, so the figure is profile-derived and labelled with its source."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        t = json.load(open(tpath))
    except Exception:
        return None, None
    ent = t.get(key)
    if isinstance(ent, dict):
        return ent.get("hbm_bytes_per_launch"), f"profiles/traffic.json[{key}] ({ent.get('source', 'rocprofv3 --pmc')})"
    return None, None


def profile_issue(key):
    """(valu_issue_busy, sustained_clock_ghz) of the dominant kernel from the committed SQ counter pass
    (profiles/traffic.json, tools/traffic_from_pmc.py) or (None, None)"""
    try:
        ent = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(key)
    except Exception:
        return None, None
    if isinstance(ent, dict):
        return ent.get("valu_issue_busy"), ent.get("sustained_clock_ghz")
    return None, None


def zero_quad_stats(X):
    """(live columns, column quads of the live columns, fraction of (sample, quad) blocks that are all zero) of a
    chunk: what the chain kernels (csrc/gen_nt_asm.py) actually execute -- dead columns are retired, an all-zero
    quad's step is 5 operations per pair instead of 6"""
    import numpy as np
    live = np.flatnonzero((X != 0).any(axis=0))
    if live.size == 0:
        return 0, 0, 1.0
    nq = (live.size + 3) // 4
    P = np.zeros((X.shape[0], nq * 4), X.dtype)
    P[:, :live.size] = X[:, live]
    z = (P.reshape(X.shape[0], nq, 4) == 0).all(axis=2)
    return int(live.size), int(nq), float(z.mean())


ARITH_MODES = ("strict", "sigma", "contracted")
ARITH_TEXT = {
    "strict": "strict (library default; everything bit-identical to the CPU oracle over whole schedules)",
    "sigma": ("sigma-contracted (S = fma(w*d,d,S) only: map / BMU indices / bmuHits / MSE / weightMap bit-identical over "
              "whole schedules, sigmaMap within 1e-5 relative: tests/test_gpu_fma_schedule.py)"),
    "contracted": ("contracted (M and S chains fused: within 1e-5 for ONE epoch from a given map; a multi-epoch schedule "
                   "leaves the reference's trajectory, profiles/r3_fma_schedule.jsonl -- throughput figure, not a parity mode)"),
}


def main():
    args = parse()
    if args.group:
        return main_group(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    cfg = args.cfg
    online = args.config == "online"

    import torch
    import torch.distributed as dist
    import vsom_amd
    from vsom_amd import capi
    import importlib
    import gen
    vdist = importlib.import_module("variational-self-organizing-maps_amd.dist")

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        backend = args.backend
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    W = H = cfg["map"]
    J = cfg["dim"]
    tr = cfg["transform"]
    sigma = cfg["sigma"]
    sharded = world > 1 and not online           # the online path is strictly sequential: replicas only

    stream = torch.cuda.Stream(device=dev)
    ctx = vsom_amd.Context(W, H, J, tr, device=local_rank)
    D = ctx.depth
    init_map = make_map(cfg, D, args.data)
    ctx.set_state(map=init_map)
    if online and args.online_search != "auto":
        ctx.set_bmu_mode({"exact": capi.BMU_EXACT, "image": capi.BMU_SHORTLIST}[args.online_search])
    eng = vdist.HipEngine(ctx, dev, stream)      # the context adopts `stream`; collectives run on it too
    trainer = vdist.ShardedBatchTrainer(eng, rank if sharded else 0, world if sharded else 1)
    is_first = not args.local

    # synthetic data.  Chunk i of the job is the same 4096-row matrix whatever N is: strong scaling
    # gives rank r rows [lo_r, hi_r) of it (generated whole from the chunk's seed, then sliced: 50 ms of host work
    # per chunk, outside every timed region); weak scaling gives every rank its own chunk-sized matrix (seed per
    # (chunk, rank)), i.e. a 4096*N-row step.  No rank holds another rank's rows before the all-gather.
    Bcfg = cfg["chunk"]

    class Split:
        def __init__(self, kind, variant=None):
            self.kind = kind
            variant = variant or args.data
            self.variant = variant
            if not sharded:
                self.Bper, self.Bglob, self.lo = Bcfg, Bcfg, 0
                self.own_host = [make_data(cfg, Bcfg, 3 + i, variant) for i in range(args.nchunks)]
            elif kind == "strong":
                lo, hi = vdist.shard_bounds(Bcfg, world, rank)
                if Bcfg % world:
                    raise SystemExit("--scaling strong needs the chunk to divide by the number of GPUs")
                self.Bper, self.Bglob, self.lo = hi - lo, Bcfg, lo
                self.own_host = [np.ascontiguousarray(make_data(cfg, Bcfg, 3 + i, variant)[lo:hi]) for i in range(args.nchunks)]
            else:
                self.Bper, self.Bglob, self.lo = Bcfg, Bcfg * world, Bcfg * rank
                self.own_host = [make_data(cfg, Bcfg, 3 + i + 1000 * rank, variant) for i in range(args.nchunks)]
            self.own = [torch.from_numpy(c).to(dev) for c in self.own_host]
            self.full = torch.empty((self.Bglob, J), dtype=torch.float32, device=dev) if sharded else None

    split = Split(args.scaling if sharded else "single")
    torch.cuda.synchronize()

    pinned = None
    if args.host_chunks != "off":
        if world != 1 or online:
            raise SystemExit("--host-chunks is a single-GPU batch measurement")
        pinned = [capi.PinnedBuffer(c.shape) for c in split.own_host]
        for pb, c in zip(pinned, split.own_host):
            pb.array[...] = c
        if args.host_chunks == "overlap":
            ctx.prefetch_chunk(pinned[0].array)

    # N = 1 batch steps are software-pipelined the way the C++ drivers are (Som::trainBatchSom in host/src/vsom_host.cpp):
    # the staging kernels of chunk i+1 (rows, live columns, int8 images: map-independent) are enqueued right after the
    # epoch of chunk i and run beside its chains; the step then starts with a commit that launches nothing
    pipelined = (not online and not sharded and args.host_chunks == "off" and not args.no_stage_ahead)
    mode = {"pipelined": pipelined}              # (switched off for the like-for-like figure `staged_in_step` below)

    def step(sp, i):
        with torch.cuda.stream(stream):
            if mode["pipelined"]:
                ctx.commit_chunk()                                   # chunk i (staged during step i-1)
                eng._bind_chunk()
                trainer.epoch(sigma, is_first)
                nxt = sp.own[(i + 1) % len(sp.own)]
                ctx.stage_next_device(nxt.data_ptr(), nxt.shape[0])
                return
            if online:
                # Som::trainBasicSom's sample loop over one staged chunk (Som.cpp:1159-1171)
                eng.load_chunk_device(sp.own[i % len(sp.own)])
                check = capi.lib().vsom_train_online_chunk(ctx._h, 0.1, float(sigma), capi.EXPONENTIAL, None)
                if check:
                    raise RuntimeError(capi.lib().vsom_last_error().decode())
                return
            if args.host_chunks == "sync":
                ctx.upload_chunk(pinned[i % len(pinned)].array)      # blocking H2D + staging
                eng._bind_chunk()
            elif args.host_chunks == "overlap":
                ctx.commit_chunk()                                   # chunk i (copied during step i-1)
                eng._bind_chunk()
            elif sharded:
                # chunk replication: every rank contributes its own rows (all-gather over xGMI)
                dist.all_gather_into_tensor(sp.full, sp.own[i % len(sp.own)])
                eng.load_chunk_device(sp.full)
            else:
                eng.load_chunk_device(sp.own[i % len(sp.own)])   # staging + lastBMU reset (DataSet.cpp:118-160)
            trainer.epoch(sigma, is_first)
            if args.host_chunks == "overlap":
                ctx.prefetch_chunk(pinned[(i + 1) % len(pinned)].array)   # H2D of chunk i+1 beside this epoch

    # HIP events of the library's kernel groups (vsom_enable_timing_of): the timed region carries the DOMINANT group's
    # events only -- every timed group puts two event records between kernels that otherwise run back to back (~5 us of
    # idle device each; twelve of them were 1.4 % of a C3 step) -- and a short second pass with all groups gives the
    # per-phase breakdown (`kernel_ms_per_step`)
    dominant = "online" if online else "update"

    def timed(sp, nwarm, nsteps, groups=(dominant,), keep_pending=False):
        if mode["pipelined"]:
            ctx.stage_next_device(sp.own[0].data_ptr(), sp.own[0].shape[0])   # the chunk of step 0
        for i in range(nwarm):
            step(sp, i)
        trainer.flush()
        torch.cuda.synchronize()
        ctx.get_timing(reset=True)
        ctx.enable_timing(True, groups=groups)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nsteps):
            step(sp, nwarm + i)
        trainer.flush()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        tm = ctx.get_timing(reset=True)
        ctx.enable_timing(False)
        if mode["pipelined"] and not keep_pending:
            # the context still holds step nsteps's chunk as "next" (a pointer into sp.own, possibly staged ahead): swap in
            # an empty one, so that the caller may free sp.own
            with torch.cuda.stream(stream):
                ctx.stage_next_device(0, 0)
                ctx.commit_chunk()
            torch.cuda.synchronize()
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, tm

    # the library applies the contracted modes to the chains that have one (vsom_hip.h, vsom_update_mode)
    mode_of = {"contracted": capi.UPDATE_FMA, "strict": capi.UPDATE_STRICT, "sigma": capi.UPDATE_FMA_SIGMA}
    ctx.set_update_mode(mode_of[args.arith])
    dt, timing = timed(split, args.warmup, args.steps)
    mse = float(ctx.get_mse())
    sl_stats = ctx.shortlist_stats() if not online else None
    onl_stats = ctx.online_search_stats() if online else None
    # like for like with rounds 1-4 and BASELINE: the same steps with every chunk staged at the start of its own step
    dt_in_step = None
    if pipelined:
        mode["pipelined"] = False
        dt_in_step, _ = timed(split, 2, args.steps)
        mode["pipelined"] = True
    bsteps = max(1, min(args.steps, 10))
    _, breakdown = timed(split, 1, bsteps, groups=capi.TIMER_NAMES)      # every group timed: the per-phase figures

    others = []
    if not args.no_other_arith and not online and capi.has_contracted(tr):   # Median / CLR have one arithmetic
        for name in ARITH_MODES:
            if name == args.arith:
                continue
            ctx.set_update_mode(mode_of[name])
            dto, tmo = timed(split, 2, args.steps)
            others.append((name, dto, tmo))
        ctx.set_update_mode(mode_of[args.arith])

    # the same step on rows of the other kinds (strict or whatever --arith says; its own chunks, its own start map)
    variants = []
    if cfg["data"] == "mnist" and not online and world == 1 and not args.no_data_variants and args.host_chunks == "off":
        for vname in DATA_VARIANTS:
            if vname == args.data:
                continue
            spv = Split("single", vname)
            torch.cuda.synchronize()
            ctx.set_state(map=make_map(cfg, D, vname))
            dtv, tmv = timed(spv, 3, args.steps)
            slv = ctx.shortlist_stats()
            _, tbv = timed(spv, 1, bsteps, groups=capi.TIMER_NAMES)
            variants.append((vname, spv, dtv, tmv, slv, tbv))
            del spv.own
        ctx.set_state(map=init_map)

    other_split = None
    if sharded and not args.no_other_scaling:
        sp2 = Split("weak" if args.scaling == "strong" else "strong")
        torch.cuda.synchronize()
        dt2, tm2 = timed(sp2, 2, args.steps)
        other_split = (sp2, dt2, tm2)

    # how many ranks really took part in a collective -- and over what: only the nccl backend is RCCL
    coll_ranks = 1
    if world > 1:
        one = torch.ones(1, dtype=torch.int32, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(one)
        coll_ranks = int(one.item())

    if rank == 0:
        N = W * H
        n_nodes_rank = vdist.shard_bounds(N, world, 0)[1] if sharded else N
        steps = args.steps

        def units_of(sp):
            return sp.Bglob if sharded else sp.Bper * world      # replicas: every rank processes its own chunk

        def roofline_for(tm, sp):
            if online:
                # HBM / Infinity-Cache bound: per sample a scan of the map (4*N*D B) + the window's
                # read-modify-write of M, S, sigma (20*k*D B), k = window nodes  
                side = min(W, 2 * int(2.5 * sigma) + 1)
                k = side * side
                bytes_sample = 4.0 * N * D + 20.0 * k * D
                t_ms, cnt = tm["online"]
                per_sample_s = t_ms / 1e3 / max(steps * sp.Bper, 1)
                ach = bytes_sample / per_sample_s / 1e9 if per_sample_s > 0 else 0.0
                image = bool(onl_stats and onl_stats["samples"] > 0)
                r = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(ach / HBM_PEAK_GBPS, 4),
                     "kernel": ("onl_fused_kernel (window update || image scan) + onl_refine_kernel (per trainSingle step)" if image
                                else "online_scan_kernel + online_window_kernel (per trainSingle step)"),
                     "avg_sample_us": round(per_sample_s * 1e6, 3),
                     "algorithmic_bytes_per_sample": bytes_sample, "window_nodes_upper_bound": k,
                     "note": "`achieved` prices SURVEY 8d's byte model (a 4*N*D scan + 20*k*D of window traffic per sample) over the "
                             "measured time per sample -- the contract's algorithmic figure, whatever the kernels really move; "
                             "window clipped at the map border moves less"}
                if image:
                    # what the image-bounded path really moves per sample (csrc/vsom_online.hip): the one-byte image of the
                    # nodes outside the window + 16 B of node scalars + the 4-byte lower bound per node; per window node M and S
                    # read and written (sigmaMap once per chunk), its image row and scalars written; the refinement reads the
                    # lower bounds and the fp32 rows of the candidates
                    ipitch = (D + 15) // 16 * 16
                    cand = onl_stats["exact_evaluations"] / max(onl_stats["samples"], 1)
                    moved = (N - k) * ipitch + N * 20.0 + k * (16.0 * D + ipitch + 20.0) + N * 4.0 + cand * 4.0 * D + 8.0 * D
                    r["moved_bytes_per_sample"] = round(moved, 0)
                    r["moved_GBps"] = round(moved / per_sample_s / 1e9, 1) if per_sample_s > 0 else 0.0
                    r["moved_frac_of_peak"] = round(moved / per_sample_s / 1e9 / HBM_PEAK_GBPS, 4) if per_sample_s > 0 else 0.0
                    r["moved_note"] = ("bytes the image-bounded search moves (model from the kernels' accesses, window at its upper "
                                       "bound): its two launches per sample are bound by their boundaries and dependent round trips, "
                                       "not by this rate")
                return r
            upd_ms, upd_cnt = tm["update"]
            upd_avg_s = upd_ms / max(upd_cnt, 1) / 1e3
            # algorithmic work of one update launch: 6 flop per (node, dim, sample) for
            # Standard / Median; CLR: 15 flop per (node, parameter pair, sample) -- the operation count of
            # Transformation.cpp:107-142 + Som.cpp:861-867 (SURVEY rounds it to 16)
            if tr == capi.CLR:
                flops = 15.0 * n_nodes_rank * (D // 2) * sp.Bglob
            else:
                flops = 6.0 * n_nodes_rank * D * sp.Bglob
            ach = flops / upd_avg_s / 1e12 if upd_avg_s > 0 else 0.0
            chain_small = bool(capi.lib().vsom_small_map_chains(ctx._h, n_nodes_rank))
            kern = {capi.STANDARD: ("update_chain3_kernel (small-map phase-2 chains)" if chain_small else
                                    "vsom_update_{std,sfma,fma}_nt4_gfx950 (phase-2 mean/sigma^2 chains, hand-scheduled)"),
                    capi.MEDIAN: ("update_chain3_kernel<median> (small-map phase-2 chains)" if chain_small else
                                  "vsom_update_med_nt4_gfx950 (phase-2 median chains, hand-scheduled)"),
                    capi.CLR: "vsom_update_clr_rp8_gfx950 (phase-2 CLR chains, hand-scheduled)"}[tr]
            r = {"bound": "valu_fp32", "achieved": round(ach, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                 "frac": round(ach / FP32_PEAK_TFLOPS, 4), "kernel": kern,
                 "note": ("fp32 VALU-bound chains priced against the fp32 dense peak the vector and matrix pipes "
                          "share (157.3 TFLOP/s counts an FMA as 2 flop); of the 6 algorithmic flop per element "
                          "strict arithmetic (no FMA) can reach 0.5 of it, sigma-contracted 0.6, contracted 0.75.  "
                          "`frac` counts ALL D columns and 6 flop per element as the work (the contract's algorithmic "
                          "figure); `frac_executed` counts what the kernel executes: only the chunk's live columns "
                          "(dead ones are retired exactly) and 5 instead of 6 flop for all-zero (sample, column quad) "
                          "blocks"),
                 "avg_launch_ms": round(upd_avg_s * 1e3, 4),
                 "algorithmic_flop_per_launch": flops}
            if tr != capi.CLR and not chain_small:
                live, nq, zfrac = zero_quad_stats(sp.own_host[0])
                per_elem = 6.0 - (zfrac if tr == capi.STANDARD else 0.0)     # Median has no cheaper zero form
                ex = per_elem * n_nodes_rank * (4.0 * nq) * sp.Bglob
                r["executed_flop_per_launch"] = ex
                r["frac_executed"] = round(ex / upd_avg_s / 1e12 / FP32_PEAK_TFLOPS, 4) if upd_avg_s > 0 else 0.0
                r["executed_basis"] = {"live_columns": live, "column_quads": nq, "zero_quad_fraction": round(zfrac, 4),
                                       "of": "rank-0 rows of chunk 0"}
            else:
                r["executed_flop_per_launch"] = flops
                r["frac_executed"] = r["frac"]
            return r

        roof = roofline_for(timing, split)
        traffic, tsrc = profile_traffic(f"{args.config}_{args.arith}")
        roof["traffic"] = traffic
        roof["traffic_source"] = tsrc
        if not online:
            busy, clk = profile_issue(f"{args.config}_{args.arith}")
            roof["valu_issue_busy"] = busy
            roof["sustained_clock_ghz"] = clk
            roof["issue_source"] = ("profiles/traffic.json (SQ_INSTS_VALU x 4 / 1024 SIMDs over GRBM_GUI_ACTIVE / 8; the clock "
                                    "is those cycles over the launch's duration under the counter pass)")
        arith = "n/a (online path has no contracted mode)" if online else ARITH_TEXT[args.arith]
        if not online and args.arith != "strict" and not capi.has_contracted(tr):
            arith = "strict (this transformation has ONE arithmetic, bit-identical in every mode)"
        live, cols = gen.column_occupancy(split.own_host[0])
        out = {
            "metric": METRIC if args.config == "c3" else "training samples/sec per epoch (BMU+update)",
            "value": round(steps * units_of(split) / dt, 3),
            "unit": "samples/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(dt / steps * 1e3, 4),
            "higher_is_better": True,
            # the split this command line applies as N grows (the driver compares the per-N lines of ONE command): strong for
            # the batch path (BASELINE config 3: the 4096-row chunk sharded), weak for the online path (replicas)
            "scaling": "weak" if online else args.scaling,
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "update_arithmetic": arith,
            "staging": ("chunk i+1 staged beside the chains of chunk i (vsom_stage_next_device + vsom_commit_chunk; every step "
                        "still stages exactly one chunk)" if pipelined else "every chunk staged at the start of its own step"),
            "staged_in_step": ({"value": round(steps * units_of(split) / dt_in_step, 3),
                                "ms_per_step": round(dt_in_step / steps * 1e3, 4),
                                "note": "the same steps with every chunk staged at the start of its own step (--no-stage-ahead: "
                                        "how rounds 1-4 and the BENCH_r01-r04 lines were timed)"} if dt_in_step else None),
            "backend": ({"nccl": "nccl (RCCL)", "gloo": "gloo (CPU rehearsal of the N > 1 flow, not RCCL)"}[backend]
                        if backend else "none (1 GPU)"),
            "collective_ranks": coll_ranks,
            "rccl_ranks": coll_ranks if backend == "nccl" else (1 if world == 1 else 0),
            "config": {"workload": (f"{cfg['name']}, "
                                    + ("trainBasicSom sample loop (trainSingle per sample)" if online else
                                       f"trainBatchSomEpoch({'findBmu' if is_first else 'findLocalBmu'} + update)")
                                    + f", chunk B={split.Bper}/GPU ({split.Bglob if sharded else split.Bper} per step and "
                                    + ("replica" if not sharded and world > 1 else "job") + f"), sigma={sigma}"),
                       "baseline_config": args.config,
                       "map": [W, H], "dim": J, "depth": D, "chunk_per_gpu": split.Bper,
                       "chunk_total": split.Bglob if sharded else split.Bper * world,
                       "column_occupancy": {"live": live, "columns": cols,
                                            "note": "columns with a non-zero in the rank-0 rows of chunk 0 (tests/gen.py: "
                                                    "661 of 784 per 4096 rows, MNIST-like; real MNIST when VSOM_MNIST_DIR is set)"},
                       "parallelism": (f"{world} GPU: phase 1 sample-sharded, phase 2 node-sharded, all-gathers of "
                                       "X / lastBMU / sqres / map rows inside the step" if sharded else
                                       (f"{world} independent replicas (the online path is sequential in samples)"
                                        if world > 1 else "1 GPU"))},
            "roofline": roof,
            "kernel_ms_per_step": {k: round(v[0] / bsteps, 4) for k, v in breakdown.items()},
            "kernel_ms_note": (f"from a second pass of {bsteps} steps with every kernel group timed (its event records cost "
                               "idle device time, so its steps are slower than `ms_per_step`); the timed region of `value` "
                               f"carries the events of the dominant group ('{dominant}') only"),
            "mse_last": mse,
            "bmu_shortlist_last": sl_stats,
        }
        if onl_stats is not None:
            n_s = max(onl_stats["samples"], 1)
            out["online_search"] = {
                "mode": args.online_search,
                "through_the_image": onl_stats["samples"] > 0,
                "samples": onl_stats["samples"],
                "exact_evaluations_per_sample": round(onl_stats["exact_evaluations"] / n_s, 2),
                "refine_workgroups_per_sample": round(onl_stats["refine_workgroups"] / n_s, 2)}
        if others:
            out["other_arithmetics"] = []
            for oname, dto, tmo in others:
                ro = roofline_for(tmo, split)
                out["other_arithmetics"].append({
                    "arithmetic": oname, "note": ARITH_TEXT[oname],
                    "value": round(steps * units_of(split) / dto, 3), "ms_per_step": round(dto / steps * 1e3, 4),
                    "update_avg_launch_ms": ro.get("avg_launch_ms"), "update_achieved_tflops": ro.get("achieved"),
                    "update_frac_of_peak": ro.get("frac")})
        if variants or cfg["data"] == "mnist":
            out["data"] = "synthetic"
            out["data_kind"] = {"name": split.variant, "note": DATA_TEXT[split.variant]}
        if variants:
            out["data_variants"] = []
            for vname, spv, dtv, tmv, slv, tbv in variants:
                rv = roofline_for(tmv, spv)
                lv, cv = gen.column_occupancy(spv.own_host[0])
                out["data_variants"].append({
                    "data": vname, "note": DATA_TEXT[vname],
                    "value": round(steps * units_of(spv) / dtv, 3), "ms_per_step": round(dtv / steps * 1e3, 4),
                    "vs_headline": round(dt / dtv, 4),
                    "bmu_ms": round(tbv["bmu"][0] / bsteps, 4), "update_ms": round(tmv["update"][0] / steps, 4),
                    "kernel_ms_per_step": {k: round(v[0] / bsteps, 4) for k, v in tbv.items()},
                    "live_columns": lv,
                    "roofline": {k: rv.get(k) for k in ("achieved", "frac", "frac_executed", "avg_launch_ms",
                                                         "executed_basis")},
                    "bmu_shortlist_last": slv})
        if other_split is not None:
            sp2, dt2, tm2 = other_split
            out[sp2.kind + "_scaling"] = {
                "scaling": sp2.kind, "value": round(steps * units_of(sp2) / dt2, 3), "unit": "samples/s",
                "ms_per_step": round(dt2 / steps * 1e3, 4), "chunk_per_gpu": sp2.Bper, "chunk_total": sp2.Bglob,
                "update_ms_per_step": round(tm2[dominant][0] / steps, 4),
                "note": ("every rank owns a full 4096-row chunk: one step is a 4096*N-row trainBatchSomEpoch, a different "
                         "algorithmic step from BASELINE config 3 (each chunk overwrites the map, Som.cpp:870)"
                         if sp2.kind == "weak" else "the 4096-row chunk of BASELINE config 3 shared out over the ranks")}
        if not args.no_cpu and world == 1:   # the CPU leg runs at N=1 only (contract)
            out["cpu_baseline"] = cpu_baseline(args, split.own_host[0][:split.Bper], init_map, online=online)
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def main_group(args):
    """--group: ONE process, N devices, the C ABI's vsom_group_* entry points (csrc/vsom_group.hip): what the C++
    mirror's Som::train(BatchMap) runs when VSOM_DEVICES names several GPUs.  Strong split by construction (the
    group shards the chunk it is handed); member r's own rows are resident on device r before the timed region
    and vsom_group_set_chunk_device all-gathers them inside it."""
    import torch
    import vsom_amd
    from vsom_amd import capi
    import gen
    cfg = args.cfg
    if args.config == "online":
        raise SystemExit("--group: the online path does not shard (replicas only)")
    n = max(1, args.gpus)
    visible = capi.device_count()
    if visible < 1:
        raise SystemExit("bench.py needs a GPU (no CPU fallback in the product path)")
    devices = [0] * n if args.share_device else list(range(n))
    if not args.share_device and n > visible:
        raise SystemExit(f"--group --gpus {n}: only {visible} device(s) visible (add --share-device to rehearse on one)")
    W = H = cfg["map"]
    J, tr, sigma, B = cfg["dim"], cfg["transform"], cfg["sigma"], cfg["chunk"]
    grp = capi.Group(W, H, J, tr, devices=devices)
    D = grp.depth
    init_map = make_map(cfg, D)
    grp.set_state(map=init_map)
    c0 = grp.member(0)
    mode_of = {"contracted": capi.UPDATE_FMA, "strict": capi.UPDATE_STRICT, "sigma": capi.UPDATE_FMA_SIGMA}
    grp.set_update_mode(mode_of[args.arith])
    chunks = [make_data(cfg, B, seed=3 + i) for i in range(args.nchunks)]
    bounds = [((B * r) // n, (B * (r + 1)) // n) for r in range(n)]
    own = [[torch.from_numpy(np.ascontiguousarray(c[lo:hi])).to(torch.device("cuda", devices[r]))
            for r, (lo, hi) in enumerate(bounds)] for c in chunks]
    for d in set(devices):
        torch.cuda.synchronize(d)
    is_first = not args.local

    def step(i):
        rows = own[i % len(own)]
        grp.set_chunk_device([t.data_ptr() for t in rows], B)
        grp.batch_epoch_async(sigma, is_first)

    def timed(nwarm, nsteps):
        for i in range(nwarm):
            step(i)
        grp.synchronize()
        c0.get_timing(reset=True)
        c0.enable_timing(True)
        t0 = time.perf_counter()
        for i in range(nsteps):
            step(nwarm + i)
        grp.synchronize()
        dt = time.perf_counter() - t0
        tm = c0.get_timing(reset=True)
        c0.enable_timing(False)
        return dt, tm

    dt, tm = timed(args.warmup, args.steps)
    steps = args.steps
    upd_ms, upd_cnt = tm["update"]
    upd_avg_s = upd_ms / max(upd_cnt, 1) / 1e3
    n_nodes0 = (W * H) // n
    flops = (15.0 * n_nodes0 * (D // 2) * B) if tr == capi.CLR else (6.0 * n_nodes0 * D * B)
    ach = flops / upd_avg_s / 1e12 if upd_avg_s > 0 else 0.0
    live, cols = gen.column_occupancy(chunks[0])
    out = {
        "metric": METRIC if args.config == "c3" else "training samples/sec per epoch (BMU+update)",
        "value": round(steps * B / dt, 3), "unit": "samples/s", "n_gpus": n, "steps": steps, "warmup": args.warmup,
        "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic", "update_arithmetic": ARITH_TEXT[args.arith],
        "front_end": "vsom_group_* (one process, csrc/vsom_group.hip)",
        "backend": {"rccl": "rccl (ncclCommInitAll, in-process)", "peer": "peer copies (hipMemcpyAsync device-to-device"
                    + (", every member on device 0: rehearsal" if args.share_device else "") + ")"}[grp.transport],
        "rccl_ranks": n if grp.transport == "rccl" else 0,
        "config": {"workload": f"{cfg['name']}, trainBatchSomEpoch({'findBmu' if is_first else 'findLocalBmu'} + update), "
                               f"chunk B={B} per step shared out over {n} member(s), sigma={sigma}",
                   "baseline_config": args.config, "map": [W, H], "dim": J, "depth": D, "chunk_per_gpu": B // n,
                   "chunk_total": B, "column_occupancy": {"live": live, "columns": cols},
                   "parallelism": f"{n} member(s): phase 1 sample-sharded, phase 2 node-sharded, gathers inside the step"},
        "roofline": {"bound": "valu_fp32", "achieved": round(ach, 3), "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(ach / FP32_PEAK_TFLOPS, 4), "kernel": "member 0's phase-2 chain kernel",
                     "avg_launch_ms": round(upd_avg_s * 1e3, 4), "algorithmic_flop_per_launch": flops, "traffic": None},
        "kernel_ms_per_step_member0": {k: round(v[0] / steps, 4) for k, v in tm.items()},
        "mse_last": float(grp.get_mse()),
    }
    print(json.dumps(out), flush=True)
    grp.close()


if __name__ == "__main__":
    main()
