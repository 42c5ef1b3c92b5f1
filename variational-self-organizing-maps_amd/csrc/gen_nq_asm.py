#!/usr/bin/env python3
"""Phase-2 chain kernels for node shards and mid-sized maps: lane = (node, four dims), operands through LDS.

Som::trainBatchSomEpoch phase 2 (Som.cpp:840-870), Standard (strict / sigma-contracted / contracted) and
StandardMedianEstimator, same per-element operation sequence as the lane = node kernels of gen_update_asm.py
(so the same bits).  What differs is the decomposition.  lane = node with 14 dims per lane gives
ceil(N/64) * ceil(D/14) wavefronts: 3.5 per SIMD on a 64x64x784 map (BASELINE config 2), 1.75 on the
2048-node shard of config 3's 8-GPU split -- whole wavefronts that cannot be balanced over 1024 SIMDs (half of
them run one more than the other half) and too few to hide the scalar x round trip.  Here

    workgroup = 32 nodes x 32 columns, lane = (node, 4 consecutive columns) = two packed chains pairs,
    grid      = ceil(N/32) x ceil(columns/32) workgroups of 4 wavefronts

i.e. 4096 x 784 -> 12 800 wavefronts of small, equal work that the dispatcher spreads evenly, 6 resident per
SIMD (24 KB of LDS per workgroup, <= 80 VGPRs).  Per block of 32 samples a workgroup stages the 32 x 32 x values
(one 16-byte global load per thread, rows as they lie in memory) and the (c, w) pairs of its 32 nodes (two
16-byte loads per thread out of the pair-interleaved `cw2` array) into one of two LDS slots, one block ahead in
registers; the chains read them back as broadcasts: per sample pair two ds_read_b128 (x) and one (c, w) per
lane, three pairs in flight (counted lgkmcnt), against 24 packed VALU operations -- the LDS array is ~1/3 busy
(MI355X_MICROARCH.md: ds_read_b128 256 B/clk/CU).  One s_barrier per block.

XCD-aware grid (as gen_update_asm.py): grid.x = 8 * column blocks, grid.y = ceil(node groups / 8); workgroups
are dealt round-robin over the 8 XCDs by linear id, so id % 8 labels the XCD; an XCD holds ~8 node groups x ALL
column blocks at a time: every x block and every (c, w) block it fetches is shared by ~8 resp. ~21-25
workgroups through its L2.

Kernarg (UpdAsmArgs of vsom_update.hip, 80 bytes): xs, cw2, map, sbuf, ldx_bytes, ldn_bytes, B, nloc,
nblocks (column blocks of 32), pitch_bytes, n0, -, live record (or null: column compaction, vsom_compact.hip:
word 2 = live columns rounded up to 32), -.
Outputs: map rows (final M) and the raw S accumulator, 32 columns per block: the caller turns S into
sqrt(S/W) and re-zeroes padding columns (sigma_finalize_kernel / cc_expand_kernel) exactly as for the
lane = node kernels.
"""

import os

NW, PQ = 32, 8                  # nodes x quads (of 4 columns) per workgroup


class Cfg:
    """CT samples per staged block, RINGD sample pairs of LDS reads in flight per lane.
    CT = 32, RINGD = 3: 77 VGPRs, 24 KB of LDS -> 6 workgroups per CU (6 wavefronts per SIMD);
    CT = 16, RINGD = 2: <= 64 VGPRs, 12 KB -> 8 per SIMD (the hardware's maximum)."""

    def __init__(self, ct, ringd):
        assert ct in (16, 32)
        self.CT, self.RINGD = ct, ringd
        self.LOG = 5 if ct == 32 else 4
        self.X_XOR = ct * 128                    # x slots at 0 / CT*128 (CT rows of 128 B)
        self.C_BASE = ct * 256                   # (c,w) slots (CT/2 pair-rows of 32 nodes x 16 B) at CT*256 / CT*512
        self.C_XOR = ct * 768
        self.LDS_BYTES = ct * 768
        self.NCW = ct // 16                      # 16-byte (c,w) pieces per thread and block
        self.V_RING = V_RING
        self.V_G = V_RING + 12 * ringd           # staging: x 4, (c,w) 4 * NCW
        b = self.V_G + 4 + 4 * self.NCW
        self.V_XR, self.V_CR, self.V_XW, self.V_CW, self.V_OX, self.V_OC, self.V_OC2 = b, b + 1, b + 2, b + 3, b + 4, b + 5, b + 6
        self.NVGPR = b + (7 if self.NCW == 2 else 6)


# SGPRs
S_KARG = "s[0:1]"
S_WGX, S_WGY = "s2", "s3"       # -> node group, column block
S_XP, S_CP, S_MAP, S_SBUF = (4, 5), (6, 7), (8, 9), (10, 11)
S_LDX, S_LDN, S_B, S_NLOC, S_NB, S_PITCH, S_N0 = "s12", "s13", "s14", "s15", "s16", "s17", "s18"
S_CNT, S_TAIL, S_TMP, S_TMP2 = "s19", "s20", "s21", "s22"
S_XSTEP, S_CSTEP = "s23", "s24"
S_REC = (28, 29)
S_BIG = "s[30:31]"              # Median: both halves 2^100
S_EXEC = "s[32:33]"
S_XEXEC = "s[36:37]"            # CT = 16: the 128 threads that stage x
# VGPRs
V_TID = 0
V_M, V_S, V_D, V_T, V_U = 2, 6, 10, 14, 18        # two packed pairs each
V_RING = 22                                        # RINGD slots of {x(2j) 4, x(2j+1) 4, cw 4}
V_A = V_D                                          # epilogue address pair (v[V_D:V_D+1]), V_D+2 node


def vp(base, p):
    return f"v[{base + 2 * p}:{base + 2 * p + 1}]"


def compute(o, mode, xb, cwb):
    """one sample for the lane's two packed pairs; x in v[xb:xb+3], {c, w} in v[cwb:cwb+1].
    mode: 'std' strict | 'sfma' sigma-contracted | 'fma' contracted | 'med' Median (Som.cpp:861-867,
    Transformation.cpp:12,50) -- instruction for instruction the per-pair sequences of gen_update_asm.py"""
    cw = f"v[{cwb}:{cwb + 1}]"
    P = (0, 1)
    for p in P:   # delta = x - M
        o.append(f"\tv_pk_add_f32 {vp(V_D, p)}, {vp(xb, p)}, {vp(V_M, p)} neg_lo:[0,1] neg_hi:[0,1]")
    if mode == "med":
        for p in P:   # t = delta * 2^100
            o.append(f"\tv_pk_mul_f32 {vp(V_D, p)}, {vp(V_D, p)}, {S_BIG}")
        for p in P:   # p = [delta > 0]
            o.append(f"\tv_pk_mul_f32 {vp(V_T, p)}, {vp(V_D, p)}, {S_BIG} clamp")
        for p in P:   # n = [delta < 0]
            o.append(f"\tv_pk_mul_f32 {vp(V_U, p)}, {vp(V_D, p)}, {S_BIG} neg_lo:[1,0] neg_hi:[1,0] clamp")
        for p in P:   # M = M + c*p
            o.append(f"\tv_pk_fma_f32 {vp(V_M, p)}, {cw}, {vp(V_T, p)}, {vp(V_M, p)} op_sel_hi:[0,1,1]")
        for p in P:   # S = S + w*p
            o.append(f"\tv_pk_fma_f32 {vp(V_S, p)}, {cw}, {vp(V_T, p)}, {vp(V_S, p)} op_sel:[1,0,0]")
        for p in P:   # M = M - c*n
            o.append(f"\tv_pk_fma_f32 {vp(V_M, p)}, {cw}, {vp(V_U, p)}, {vp(V_M, p)} op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")
        for p in P:   # S = S + w*n
            o.append(f"\tv_pk_fma_f32 {vp(V_S, p)}, {cw}, {vp(V_U, p)}, {vp(V_S, p)} op_sel:[1,0,0]")
        return
    if mode == "fma":
        for p in P:   # M = c*delta + M
            o.append(f"\tv_pk_fma_f32 {vp(V_M, p)}, {cw}, {vp(V_D, p)}, {vp(V_M, p)} op_sel_hi:[0,1,1]")
    else:
        for p in P:   # t = c*delta
            o.append(f"\tv_pk_mul_f32 {vp(V_T, p)}, {cw}, {vp(V_D, p)} op_sel_hi:[0,1]")
    for p in P:       # u = w*delta
        o.append(f"\tv_pk_mul_f32 {vp(V_U, p)}, {cw}, {vp(V_D, p)} op_sel:[1,0]")
    if mode != "fma":
        for p in P:   # M = M + t                               (Som.cpp:864)
            o.append(f"\tv_pk_add_f32 {vp(V_M, p)}, {vp(V_M, p)}, {vp(V_T, p)}")
    if mode == "std":
        for p in P:   # u = u*delta
            o.append(f"\tv_pk_mul_f32 {vp(V_U, p)}, {vp(V_U, p)}, {vp(V_D, p)}")
        for p in P:   # S = S + u                               (Som.cpp:867)
            o.append(f"\tv_pk_add_f32 {vp(V_S, p)}, {vp(V_S, p)}, {vp(V_U, p)}")
    else:
        for p in P:   # S = u*delta + S
            o.append(f"\tv_pk_fma_f32 {vp(V_S, p)}, {vp(V_U, p)}, {vp(V_D, p)}, {vp(V_S, p)}")


class LdsQueue:
    """issue order of this wavefront's LDS operations (they complete in order): lgkmcnt for 'tag done'"""

    scale = 3       # (timing experiments issue fewer reads per pair)

    def __init__(self):
        self.q = []

    def push(self, tag, n=1):
        self.q += [tag] * n

    def wait_for(self, o, tag):
        last = max(i for i, t in enumerate(self.q) if t == tag)
        younger = (len(self.q) - 1 - last) * LdsQueue.scale // 3
        assert younger <= 15
        o.append(f"\ts_waitcnt lgkmcnt({younger})")
        self.q = self.q[last + 1:]


def kernel(name, mode, k):
    o = []
    E = o.append
    CT, RINGD = k.CT, k.RINGD
    V_G, V_XR, V_CR, V_XW, V_CW, V_OX, V_OC, V_OC2 = k.V_G, k.V_XR, k.V_CR, k.V_XW, k.V_CW, k.V_OX, k.V_OC, k.V_OC2
    E(f"\t.text\n\t.globl {name}\n\t.p2align 8\n\t.type {name},@function\n{name}:")
    E(f"\ts_load_dwordx8 s[4:11], {S_KARG}, 0x0")              # xs, cw2, map, sbuf
    E(f"\ts_load_dwordx4 s[12:15], {S_KARG}, 0x20")            # ldx_bytes, ldn_bytes, B, nloc
    E(f"\ts_load_dwordx2 s[16:17], {S_KARG}, 0x30")            # column blocks, pitch_bytes
    E(f"\ts_load_dword {S_N0}, {S_KARG}, 0x38")
    E(f"\ts_load_dwordx2 s[{S_REC[0]}:{S_REC[1]}], {S_KARG}, 0x40")
    E(f"\ts_and_b32 {S_TMP}, {S_WGX}, 7")                      # XCD label
    E(f"\ts_lshr_b32 {S_TMP2}, {S_WGX}, 3")                    # column block
    E(f"\ts_lshl_b32 {S_WGY}, {S_WGY}, 3")
    E(f"\ts_add_u32 {S_WGX}, {S_WGY}, {S_TMP}")                # node group = wgy*8 + xcd
    E(f"\ts_mov_b32 {S_WGY}, {S_TMP2}")
    E(f"\tv_and_b32_e32 v{V_TID}, 0x3ff, v{V_TID}")
    E(f"\ts_waitcnt lgkmcnt(0)")
    E(f"\ts_cmp_eq_u64 s[{S_REC[0]}:{S_REC[1]}], 0")
    E(f"\ts_cbranch_scc1 .L_nb_{name}")
    E(f"\ts_load_dword {S_NB}, s[{S_REC[0]}:{S_REC[1]}], 0x8")   # live columns rounded up to 32 (device value)
    E(f"\ts_waitcnt lgkmcnt(0)")
    E(f"\ts_lshr_b32 {S_NB}, {S_NB}, 5")
    E(f".L_nb_{name}:")
    E(f"\ts_cmp_ge_u32 {S_WGY}, {S_NB}")
    E(f"\ts_cbranch_scc1 .L_end_{name}")
    E(f"\ts_lshl_b32 {S_TMP}, {S_WGX}, 5")                     # first node of the workgroup
    E(f"\ts_cmp_ge_u32 {S_TMP}, {S_NLOC}")
    E(f"\ts_cbranch_scc1 .L_end_{name}")
    if mode == "med":
        E(f"\ts_mov_b32 s30, 0x71800000")                         # 2^100
        E(f"\ts_mov_b32 s31, 0x71800000")
    if CT == 16:   # x block = 16 rows x 8 pieces: threads 0..127 (wavefronts 0, 1) stage it
        E(f"\tv_readfirstlane_b32 {S_TMP}, v{V_TID}")
        E(f"\ts_cmp_lt_u32 {S_TMP}, 128")
        E(f"\ts_cselect_b32 s36, -1, 0")
        E(f"\ts_mov_b32 s37, s36")
    # per-thread addresses
    E(f"\tv_and_b32_e32 v{V_XR}, 7, v{V_TID}")
    E(f"\tv_lshlrev_b32_e32 v{V_XR}, 4, v{V_XR}")              # quad * 16: x read base (slot 0)
    E(f"\tv_lshrrev_b32_e32 v{V_CR}, 3, v{V_TID}")             # local node
    E(f"\tv_mul_lo_u32 v{V_OX}, v{V_CR}, {S_LDX}")             # staging: row tid>>3 of the block ...
    E(f"\tv_add_u32_e32 v{V_OX}, v{V_OX}, v{V_XR}")            # ... 16-byte piece tid&7
    E(f"\tv_lshlrev_b32_e32 v{V_CR}, 4, v{V_CR}")
    E(f"\tv_add_u32_e32 v{V_CR}, {k.C_BASE}, v{V_CR}")         # (c,w) read base (slot 0)
    E(f"\tv_lshlrev_b32_e32 v{V_XW}, 4, v{V_TID}")             # x write: tid * 16
    E(f"\tv_add_u32_e32 v{V_CW}, {k.C_BASE}, v{V_XW}")         # (c,w) write: pair-row tid>>5 (+8), node tid&31
    E(f"\tv_lshrrev_b32_e32 v{V_OC}, 5, v{V_TID}")
    E(f"\tv_mul_lo_u32 v{V_OC}, v{V_OC}, {S_LDN}")
    E(f"\tv_and_b32_e32 v{V_D}, 31, v{V_TID}")
    E(f"\tv_lshl_add_u32 v{V_OC}, v{V_D}, 4, v{V_OC}")
    if k.NCW == 2:
        E(f"\ts_lshl_b32 {S_TMP}, {S_LDN}, 3")
        E(f"\tv_add_u32_e32 v{V_OC2}, {S_TMP}, v{V_OC}")       # pair-rows 8..15 of the block
    # global bases: x + column block * 128 B ; cw2 + node group * 32 nodes * 16 B
    E(f"\ts_lshl_b32 {S_TMP}, {S_WGY}, 7")
    E(f"\ts_add_u32 s{S_XP[0]}, s{S_XP[0]}, {S_TMP}")
    E(f"\ts_addc_u32 s{S_XP[1]}, s{S_XP[1]}, 0")
    E(f"\ts_lshl_b32 {S_TMP}, {S_WGX}, 9")
    E(f"\ts_add_u32 s{S_CP[0]}, s{S_CP[0]}, {S_TMP}")
    E(f"\ts_addc_u32 s{S_CP[1]}, s{S_CP[1]}, 0")
    E(f"\ts_lshl_b32 {S_XSTEP}, {S_LDX}, {k.LOG}")             # CT sample rows
    E(f"\ts_lshl_b32 {S_CSTEP}, {S_LDN}, {k.LOG - 1}")         # CT/2 pair-rows
    for r in range(V_M, V_M + 8):                               # currentModel / currentModelSigma .setZero() :843-844
        E(f"\tv_mov_b32_e32 v{r}, 0")
    E(f"\ts_cmp_eq_u32 {S_B}, 0")
    E(f"\ts_cbranch_scc1 .L_store_{name}")
    E(f"\ts_add_u32 {S_CNT}, {S_B}, {CT - 1}")
    E(f"\ts_lshr_b32 {S_CNT}, {S_CNT}, {k.LOG}")
    E(f"\ts_sub_u32 {S_CNT}, {S_CNT}, 1")                      # full blocks before the last one
    E(f"\ts_lshl_b32 {S_TMP}, {S_CNT}, {k.LOG}")
    E(f"\ts_sub_u32 {S_TAIL}, {S_B}, {S_TMP}")                 # samples of the last block: 1..CT

    def xmask(on):
        if CT == 16:
            E(f"\ts_mov_b64 exec, {S_XEXEC if on else -1}")

    def gload():
        xmask(True)
        E(f"\tglobal_load_dwordx4 v[{V_G}:{V_G + 3}], v{V_OX}, s[{S_XP[0]}:{S_XP[1]}]")
        xmask(False)
        E(f"\tglobal_load_dwordx4 v[{V_G + 4}:{V_G + 7}], v{V_OC}, s[{S_CP[0]}:{S_CP[1]}]")
        if k.NCW == 2:
            E(f"\tglobal_load_dwordx4 v[{V_G + 8}:{V_G + 11}], v{V_OC2}, s[{S_CP[0]}:{S_CP[1]}]")
        E(f"\ts_add_u32 s{S_XP[0]}, s{S_XP[0]}, {S_XSTEP}")
        E(f"\ts_addc_u32 s{S_XP[1]}, s{S_XP[1]}, 0")
        E(f"\ts_add_u32 s{S_CP[0]}, s{S_CP[0]}, {S_CSTEP}")
        E(f"\ts_addc_u32 s{S_CP[1]}, s{S_CP[1]}, 0")

    def lwrite():
        xmask(True)
        E(f"\tds_write_b128 v{V_XW}, v[{V_G}:{V_G + 3}]")
        xmask(False)
        E(f"\tds_write_b128 v{V_CW}, v[{V_G + 4}:{V_G + 7}]")
        if k.NCW == 2:
            E(f"\tds_write_b128 v{V_CW}, v[{V_G + 8}:{V_G + 11}] offset:4096")

    def flip_w():
        E(f"\tv_xor_b32_e32 v{V_XW}, {k.X_XOR}, v{V_XW}")
        E(f"\tv_xor_b32_e32 v{V_CW}, {k.C_XOR}, v{V_CW}")

    EXP = os.environ.get("VSOM_GEN_NQ_EXP", "")     # timing experiments (wrong results)

    def lread(slot, jj):
        r = V_RING + 12 * slot
        E(f"\tds_read_b128 v[{r}:{r + 3}], v{V_XR} offset:{128 * (2 * jj)}")
        if EXP == "halfx":
            E(f"\ts_nop 0")
        else:
            E(f"\tds_read_b128 v[{r + 4}:{r + 7}], v{V_XR} offset:{128 * (2 * jj + 1)}")
        if EXP == "nocw":
            E(f"\ts_nop 0")
        else:
            E(f"\tds_read_b128 v[{r + 8}:{r + 11}], v{V_CR} offset:{512 * jj}")

    # ---- prologue: block 0 -> slot 0, block 1 -> registers --------------------------------------
    gload()
    E(f"\ts_waitcnt vmcnt(0)")
    lwrite()
    flip_w()
    E(f"\ts_waitcnt lgkmcnt(0)")
    E(f"\ts_cmp_eq_u32 {S_CNT}, 0")
    E(f"\ts_cbranch_scc1 .L_p1_{name}")
    gload()
    E(f".L_p1_{name}:")
    E(f"\ts_barrier")
    E(f"\ts_cmp_eq_u32 {S_CNT}, 0")
    E(f"\ts_cbranch_scc1 .L_last_{name}")
    # ---- main loop: one full block per iteration ---------------------------------------------
    E(f"\t.p2align 6\n.L_loop_{name}:")
    q = LdsQueue()
    E(f"\ts_waitcnt vmcnt(0)")                                # block b+1 landed in the staging registers
    lwrite()                                                    # -> the slot block b-1 was read from (LDS operations
    for jj in range(RINGD):                                     #    complete in order: the reads below wait for them)
        lread(jj, jj)
        q.push(("r", jj), 3)
    for jj in range(CT // 2):
        q.wait_for(o, ("r", jj))
        if jj == 0:
            # the writes have left the staging registers: block b+2 -> registers (if there is one)
            E(f"\ts_cmp_lt_u32 {S_CNT}, 2")
            E(f"\ts_cbranch_scc1 .L_nl_{name}")
            gload()
            E(f".L_nl_{name}:")
        r = V_RING + 12 * (jj % RINGD)
        compute(o, mode, r, r + 8)
        compute(o, mode, r + 4, r + 10)
        if jj + RINGD < CT // 2:
            lread(jj % RINGD, jj + RINGD)
            q.push(("r", jj + RINGD), 3)
    flip_w()
    E(f"\tv_xor_b32_e32 v{V_XR}, {k.X_XOR}, v{V_XR}")
    E(f"\tv_xor_b32_e32 v{V_CR}, {k.C_XOR}, v{V_CR}")
    E(f"\ts_waitcnt lgkmcnt(0)")
    E(f"\ts_barrier")                                          # slot b+1 written by all, slot b read by all
    E(f"\ts_sub_u32 {S_CNT}, {S_CNT}, 1")
    E(f"\ts_cmp_lg_u32 {S_CNT}, 0")
    E(f"\ts_cbranch_scc1 .L_loop_{name}")
    # ---- last block: 1..CT samples, no staging --------------------------------------------------
    E(f".L_last_{name}:")
    lread(0, 0)
    for jj in range(CT // 2):
        E(f"\ts_cmp_le_u32 {S_TAIL}, {2 * jj}")
        E(f"\ts_cbranch_scc1 .L_store_{name}")
        if jj + 1 < CT // 2:
            lread((jj + 1) % 2, jj + 1)
            E(f"\ts_waitcnt lgkmcnt(3)")
        else:
            E(f"\ts_waitcnt lgkmcnt(0)")
        r = V_RING + 12 * (jj % 2)
        compute(o, mode, r, r + 8)
        E(f"\ts_cmp_le_u32 {S_TAIL}, {2 * jj + 1}")
        E(f"\ts_cbranch_scc1 .L_store_{name}")
        compute(o, mode, r + 4, r + 10)
    # ---- epilogue: map row <- M (Som.cpp:870), sigma buffer <- raw S ------------------------------
    E(f".L_store_{name}:")
    E(f"\ts_waitcnt vmcnt(0) lgkmcnt(0)")
    VN = f"v{V_A + 2}"
    VA = f"v[{V_A}:{V_A + 1}]"
    E(f"\tv_lshrrev_b32_e32 {VN}, 3, v{V_TID}")
    E(f"\ts_lshl_b32 {S_TMP}, {S_WGX}, 5")
    E(f"\tv_add_u32_e32 {VN}, {S_TMP}, {VN}")                   # local node index
    E(f"\tv_cmp_gt_u32_e32 vcc, {S_NLOC}, {VN}")
    E(f"\ts_and_saveexec_b64 {S_EXEC}, vcc")
    E(f"\ts_cbranch_execz .L_end_{name}")
    E(f"\tv_add_u32_e32 {VN}, {S_N0}, {VN}")                    # global node index
    E(f"\tv_and_b32_e32 v{V_XR}, 7, v{V_TID}")
    E(f"\tv_lshlrev_b32_e32 v{V_XR}, 4, v{V_XR}")
    E(f"\ts_lshl_b32 {S_TMP}, {S_WGY}, 7")                     # column block * 128 B
    for base, tag in ((S_MAP, V_M), (S_SBUF, V_S)):
        E(f"\ts_add_u32 {S_TMP2}, s{base[0]}, {S_TMP}")
        E(f"\ts_addc_u32 s34, s{base[1]}, 0")
        E(f"\tv_add_co_u32_e32 v{V_A}, vcc, {S_TMP2}, v{V_XR}")
        E(f"\tv_mov_b32_e32 v{V_A + 1}, s34")
        E(f"\tv_addc_co_u32_e32 v{V_A + 1}, vcc, 0, v{V_A + 1}, vcc")
        E(f"\tv_mad_u64_u32 {VA}, s[34:35], {VN}, {S_PITCH}, {VA}")
        E(f"\tglobal_store_dwordx4 {VA}, v[{tag}:{tag + 3}], off")
    E(f".L_end_{name}:")
    E(f"\ts_endpgm")
    E(f".L_func_end_{name}:")
    E(f"\t.size {name}, .L_func_end_{name}-{name}")
    return "\n".join(o)


MODES = ("std", "fma", "sfma", "med")


def emit():
    """[(name, text, vgprs, kernarg bytes, lds bytes, dx10_clamp)] for gen_update_asm.main()"""
    k = Cfg(int(os.environ.get("VSOM_GEN_NQ_CT", "32")), int(os.environ.get("VSOM_GEN_NQ_RING", "3")))
    if os.environ.get("VSOM_GEN_NQ_EXP", "") in ("halfx", "nocw"):
        LdsQueue.scale = 2
    out = []
    for m in MODES:
        name = f"vsom_update_{m}_nq32_gfx950"
        out.append((name, kernel(name, m, k), k.NVGPR, 80, k.LDS_BYTES, 0 if m == "med" else 1))
    return out
