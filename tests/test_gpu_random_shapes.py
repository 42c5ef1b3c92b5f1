"""GPU: seeded random sweep over map shapes, chunk lengths, transformations and sigmas -- two batch
epochs (full search, then local search) and a short online chunk per case, every output compared
bit for bit (NaN == NaN) with the oracle.  Small sigmas on wide maps make the float neighbourhood
weight underflow, so rows with W = 0 -> c = 0/0 = NaN are part of the sweep."""
import os

import numpy as np
import pytest

import gen
import vsom_amd
from vsom_amd import capi
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu


def _same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype.kind == "f":
        return bool(((a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))).all())
    return bool((a == b).all())


def _cases(n, seed):
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        tr = [po.STANDARD, po.MEDIAN, po.CLR][i % 3]
        wide = os.environ.get("VSOM_SWEEP_WIDE") == "1"
        W, H = int(rs.randint(2, 72 if wide else 40)), int(rs.randint(2, 72 if wide else 40))
        J = int(rs.randint(2, 12 if wide else 9)) if tr == po.CLR else int(rs.choice(
            [1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 33, 64, 100] + ([65, 128, 200] if wide else [])))
        B = int(rs.choice([1, 2, 3, 15, 16, 17, 63, 64, 65, 127, 200, 257] + ([300, 511, 600] if wide else [])))
        sigma = float(rs.choice([1.5, 2.0, 3.7, 8.0, 20.0]))
        out.append((f"{i}_{['std', 'med', 'clr'][tr]}_{W}x{H}x{J}_B{B}_s{sigma}", tr, W, H, J, B, sigma, int(rs.randint(1, 1 << 30))))
    return out


# VSOM_SWEEP_N / VSOM_SWEEP_SEED widen the sweep for occasional long runs (default: 36 cases)
CASES = _cases(int(os.environ.get("VSOM_SWEEP_N", "36")), int(os.environ.get("VSOM_SWEEP_SEED", "20240611"))) + [
    # shapes that reach the hand-scheduled kernels (enough wavefronts) with ragged chunks / dims
    ("asm_rd14_40x33x784_B65", po.STANDARD, 40, 33, 784, 65, 12.0, 11),
    ("asm_rd16_tail_37x35x794_B33", po.STANDARD, 37, 35, 794, 33, 9.0, 12),
    ("asm_rd14_50x50x112_B77", po.STANDARD, 50, 50, 112, 77, 6.0, 13),
    ("asm_rd16_64x64x100_B130", po.STANDARD, 64, 64, 100, 130, 25.0, 14),
    ("lane_node_median_48x48x200_B70", po.MEDIAN, 48, 48, 200, 70, 14.0, 15),
    ("asm_clr_J9_30x30_B55", po.CLR, 30, 30, 9, 55, 9.0, 16),
    ("asm_clr_J12_20x21_B129", po.CLR, 20, 21, 12, 129, 2.5, 17),
]


@pytest.mark.parametrize("name,tr,W,H,J,B,sigma,seed", CASES, ids=[c[0] for c in CASES])
def test_random_shape(name, tr, W, H, J, B, sigma, seed):
    D = po.length(tr, J)
    rs = np.random.RandomState(seed)
    X = (rs.randn(B, J) * rs.choice([0.1, 1.0, 50.0])).astype(np.float32)
    X[rs.rand(B, J) < 0.1] = 0.0                       # exact zeros (sign(0), -0 paths)
    init = gen.random_map(W * H, D, seed=seed % 1000)
    ctx = vsom_amd.Context(W, H, J, tr)
    orc = po.OracleSom(W, H, J, tr)
    ctx.set_state(map=init)
    orc.set_state(map=init)
    for first in (True, False):
        lb = np.zeros(B, np.uint64)
        mse_o = orc.batch_epoch(X, lb, sigma, first)
        ctx.upload_chunk(X)
        mse_g = ctx.batch_epoch(sigma, first)
        assert _same(ctx.get_last_bmu(), lb), (name, first, "lastBMU")
        assert _same(np.float32(mse_g), np.float32(mse_o)), (name, first, "mse")
        st = ctx.get_state()
        for k, ref in (("map", orc.map), ("sigma", orc.sigma), ("weight", orc.weight), ("hits", orc.hits)):
            assert _same(st[k], ref), (name, first, k)
    # a few online steps on top (window + post fused launch; sigma <= 1 single launch)
    clean = gen.random_map(W * H, D, seed=7)          # the batch result may hold NaN rows: restart clean
    ctx.set_state(map=clean, sigma=np.zeros_like(clean), S=np.zeros_like(clean), weight=np.zeros(W * H, np.float32),
                  hits=np.zeros(W * H, np.uint64))
    orc.set_state(map=clean, sigma=np.zeros_like(clean), S=np.zeros_like(clean), weight=np.zeros(W * H, np.float32),
                  hits=np.zeros(W * H, np.uint64))
    nb = min(B, 12)
    # maps of at most 4096 values (Standard / Median) train the chunk in ONE launch under VSOM_BMU_AUTO
    # (online_tiny_chunk_kernel); VSOM_BMU_EXACT keeps the per-sample kernels covered on those shapes too
    one_launch = tr != capi.CLR and W * H * D <= 4096 and W * H <= 1024 and D <= 512
    for mode in ((capi.BMU_AUTO, capi.BMU_EXACT) if one_launch else (capi.BMU_AUTO,)):
        ctx.set_bmu_mode(mode)
        for sg, fn in ((max(sigma, 1.2), capi.EXPONENTIAL), (1.0, capi.INVERSE_PROPORTIONAL)):
            lb = np.zeros(nb, np.uint64)
            mse_o = orc.train_online_chunk(X[:nb], lb, 0.05, sg, fn)
            ctx.upload_chunk(X[:nb])
            mse_g = ctx.train_online_chunk(0.05, sg, fn)
            assert _same(ctx.get_last_bmu(), lb), (name, sg, mode, "online lastBMU")
            assert _same(np.float32(mse_g), np.float32(mse_o)), (name, sg, mode, "online mse")
            st = ctx.get_state()
            for k, ref in (("map", orc.map), ("sigma", orc.sigma), ("S", orc.S), ("weight", orc.weight), ("hits", orc.hits)):
                assert _same(st[k], ref), (name, sg, mode, "online " + k)
    ctx.set_bmu_mode(capi.BMU_AUTO)
    # and the same chunk once more through the image-bounded search of the chunk loop (csrc/vsom_online.hip; forced: these
    # maps are far below the size VSOM_BMU_AUTO takes it for): random shapes, scales 0.1 / 1 / 50, exact zeros
    if tr != capi.CLR and D <= 1024:
        ctx.set_bmu_mode(capi.BMU_SHORTLIST)
        sg = max(sigma, 1.2)
        lb = np.zeros(nb, np.uint64)
        mse_o = orc.train_online_chunk(X[:nb], lb, 0.05, sg, capi.EXPONENTIAL)
        ctx.upload_chunk(X[:nb])
        mse_g = ctx.train_online_chunk(0.05, sg, capi.EXPONENTIAL)
        assert ctx.online_search_stats(reset=True)["samples"] == nb, name
        assert _same(ctx.get_last_bmu(), lb), (name, sg, "image-bounded online lastBMU")
        assert _same(np.float32(mse_g), np.float32(mse_o)), (name, sg, "image-bounded online mse")
        st = ctx.get_state()
        for k, ref in (("map", orc.map), ("sigma", orc.sigma), ("S", orc.S), ("weight", orc.weight), ("hits", orc.hits)):
            assert _same(st[k], ref), (name, sg, "image-bounded online " + k)
    ctx.close()


def _asm_cases(n, seed):
    """Standard-transformation shapes large enough for the hand-scheduled update kernels, with random
    depths (even, odd, multiples of 14/16 and not: every column plan and the padded last slice), random
    chunk lengths (all residues mod 8: the ring tail) and sharded second phases."""
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        W, H = int(rs.randint(34, 90)), int(rs.randint(34, 90))
        J = int(rs.choice([rs.randint(17, 130), rs.randint(130, 900)]))
        B = int(rs.choice([rs.randint(1, 40), rs.randint(40, 300)]))
        sigma = float(rs.choice([2.0, 6.0, 15.0, 40.0]))
        out.append((f"asm{i}_{W}x{H}x{J}_B{B}_s{sigma}", W, H, J, B, sigma, int(rs.randint(1, 1 << 30))))
    return out


# VSOM_ASM_SWEEP_N widens this one for occasional long runs (default: 8 cases)
ASM_CASES = _asm_cases(int(os.environ.get("VSOM_ASM_SWEEP_N", "8")), int(os.environ.get("VSOM_SWEEP_SEED", "20240611")))


@pytest.mark.parametrize("tr", [po.STANDARD, po.MEDIAN], ids=["std", "median"])
@pytest.mark.parametrize("name,W,H,J,B,sigma,seed", ASM_CASES, ids=[c[0] for c in ASM_CASES])
def test_random_shape_assembly_update(name, W, H, J, B, sigma, seed, tr):
    rs = np.random.RandomState(seed)
    X = (rs.randn(B, J) * rs.choice([0.1, 1.0, 50.0])).astype(np.float32)
    X[rs.rand(B, J) < 0.1] = 0.0
    init = gen.random_map(W * H, J, seed=seed % 1000)
    if tr == po.MEDIAN:
        # what the packed sign of the Median kernels (clamped multiplications, gen_update_asm.py) must get
        # right: -0, denormal and huge differences, +-inf and NaN samples, x == M exactly
        X[rs.rand(B, J) < 0.02] = -0.0
        X[rs.rand(B, J) < 0.01] = np.float32(1e-42)
        X[rs.rand(B, J) < 0.01] = np.float32(-3e-45)
        X[rs.rand(B, J) < 0.01] = np.float32(3e38)
        X[rs.rand(B, J) < 0.002] = np.inf
        X[rs.rand(B, J) < 0.002] = -np.inf
        X[rs.rand(B, J) < 0.002] = np.nan
        init[rs.rand(*init.shape) < 0.05] = 0.0
    # columns that are zero in EVERY row (two cases in three): the library retires their chains and their share of
    # the search's contraction exactly (csrc/vsom_compact.hip; threshold lowered to 1 row: these chunks are short)
    frac = [0.0, 0.25, 0.6][seed % 3]
    X[:, rs.rand(J) < frac] = 0.0
    # ... and runs of zeros inside single rows: (sample, 14-column slice) blocks that are entirely zero take the
    # chains' zero-quad form (gen_nt_asm.py, compute_zero), -0.0 included
    for r in np.flatnonzero(rs.rand(B) < 0.4):
        c0 = int(rs.randint(0, J))
        X[r, c0:c0 + int(rs.randint(14, 90))] = 0.0 if rs.rand() < 0.8 else -0.0
    if B > 3:
        X[int(rs.randint(0, B))] = 0.0                       # a whole row of zeros
    ctx = vsom_amd.Context(W, H, J, tr)
    ctx.set_column_compaction(1)
    orc = po.OracleSom(W, H, J, tr)
    ctx.set_state(map=init)
    orc.set_state(map=init)
    lb = np.zeros(B, np.uint64)
    mse_o = orc.batch_epoch(X, lb, sigma, True)
    ctx.upload_chunk(X)
    mse_g = ctx.batch_epoch(sigma, True)
    got = ctx.get_last_bmu()
    if not _same(got, lb):      # say which samples, what they hold and what a second search returns
        bad = np.nonzero(got != lb)[0]
        again = ctx.bmu_batch()[0]
        ctx.set_bmu_mode(vsom_amd.capi.BMU_EXACT)
        exact = ctx.bmu_batch()[0]
        info = [(int(s_), int(lb[s_]), int(got[s_]), int(again[s_]), int(exact[s_]), int(np.isnan(X[s_]).sum()),
                 int(np.isinf(X[s_]).sum()), int((np.abs(X[s_]) > 1e38).sum())) for s_ in bad[:8]]
        raise AssertionError((name, "lastBMU", len(bad), "sample/oracle/epoch/again/exact/nan/inf/big", info,
                              ctx.shortlist_stats()))
    assert _same(np.float32(mse_g), np.float32(mse_o)), (name, "mse")
    st = ctx.get_state()
    for k, ref in (("map", orc.map), ("sigma", orc.sigma), ("weight", orc.weight), ("hits", orc.hits)):
        assert _same(st[k], ref), (name, k)
    # the same epoch again as a search + two node shards of the second phase (the multi-GPU entry points)
    ctx.set_state(map=init, hits=np.zeros(W * H, np.uint64))
    ctx.upload_chunk(X)
    ctx.batch_phase1_async(0, B, True)
    ctx.batch_finish_async()
    cut = int(rs.randint(1, W * H))
    ctx.batch_phase2_async(sigma, cut, W * H)
    ctx.batch_phase2_async(sigma, 0, cut)
    ctx.synchronize()
    st = ctx.get_state()
    for k, ref in (("map", orc.map), ("sigma", orc.sigma), ("weight", orc.weight), ("hits", orc.hits)):
        assert _same(st[k], ref), (name, "sharded " + k)
    ctx.close()


@pytest.mark.parametrize("tr", [po.STANDARD, po.MEDIAN], ids=["std", "median"])
@pytest.mark.parametrize("B", [1, 2, 3, 7, 8, 9])
def test_shortest_chunks_on_the_assembly_update_path(B, tr):
    """B = 1, 2: the assembly kernels' ring of (c,w) loads and the x-row read-ahead reach past the chunk
    from the first instruction on (VSOM_ROW_PAD spare rows, ceil(B/2)+8 pair rows); 48x48x128 is past the
    chain-kernel threshold, so lane = node assembly kernels run."""
    W = H = 48
    J = 128
    rs = np.random.RandomState(100 + B)
    X = rs.randn(B, J).astype(np.float32)
    init = gen.random_map(W * H, J, seed=7)
    ctx = vsom_amd.Context(W, H, J, tr)
    orc = po.OracleSom(W, H, J, tr)
    ctx.set_state(map=init)
    orc.set_state(map=init)
    # a longer chunk first: the spare rows behind the short one then hold stale samples, not zeros
    ctx.upload_chunk(rs.randn(64, J).astype(np.float32))
    ctx.upload_chunk(X)
    lb = np.zeros(B, np.uint64)
    mse_o = orc.batch_epoch(X, lb, 6.0, True)
    mse_g = ctx.batch_epoch(6.0, True)
    assert _same(ctx.get_last_bmu(), lb) and _same(np.float32(mse_g), np.float32(mse_o))
    st = ctx.get_state()
    for k, ref in (("map", orc.map), ("sigma", orc.sigma), ("weight", orc.weight), ("hits", orc.hits)):
        assert _same(st[k], ref), (B, k)
    ctx.close()
