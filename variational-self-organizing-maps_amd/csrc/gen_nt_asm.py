#!/usr/bin/env python3
"""Phase-2 chain kernels, lane = node with FOUR dims per wavefront: x from scalar loads of a transposed chunk,
(c, w) shared by the workgroup through LDS.

Som::trainBatchSomEpoch phase 2 (Som.cpp:840-870), Standard (strict / sigma-contracted / contracted) and
StandardMedianEstimator; per element the same operation sequence as gen_update_asm.py's kernels (same bits).

Decomposition.  lane = node (64 nodes per wavefront) as in gen_update_asm.py, but a wavefront owns ONE column quad
(4 dims = two packed pairs) instead of 14-16 dims, and a workgroup is 8 wavefronts = 8 consecutive quads (32
columns) of the same 64 nodes:
  * 128x128x784 -> 50 176 wavefronts, 64x64x784 -> 12 544, a 2048-node shard 6 272: small equal pieces the
    dispatcher balances over the 1024 SIMDs, six resident per SIMD (~50 VGPRs, 32 KB of LDS per workgroup).
  * x is a wave-uniform operand: the chunk is stored transposed, Xq[quad][sample] = float4 (vsom_xq.hip), so ONE
    s_load_dwordx16 brings the wavefront's four values of FOUR consecutive samples; they feed v_pk_* as SGPR
    operands -- no vector-memory or LDS traffic for x at all.
  * (c, w) is per lane (node): per block of 32 samples the workgroup stages the 16 pair-rows x 64 nodes of
    `cw2` (two 16-byte global loads per thread) into one of two LDS slots, one block ahead in registers; every
    wavefront reads {c_j, w_j, c_j+1, w_j+1} with one ds_read_b128 per sample pair (the LDS array is ~1/6 busy
    against ~1/2 for a lane = (node, quad) decomposition that reads x from LDS as well).  One s_barrier per block.
  * all-zero quads: bit j of zq[quad][j/32] says that the four values of sample j are all +-0; such a step is
    delta = -M, i.e. t = c*M ; u = w*M ; u = u*M ; M = M - t ; S = S + u -- 5 packed operations per pair instead
    of 6, bit-identical (gen_update_asm.py, compute_zero_x).  ~70 % of the (sample, quad) blocks of an MNIST chunk.
    The wavefront branches on a scalar bit per sample.
Scalar loads return out of order, so every wait is lgkmcnt(0): at the top of each group of 4 samples the
wavefront waits for everything it issued one group earlier (x of this group, (c, w) of this group's two pairs),
issues the next group's loads and computes 40-48 packed operations.

XCD-aware grid: grid.x = 8 * column blocks, grid.y = ceil(node groups / 8) (gen_update_asm.py).

Kernarg (UpdAsmArgs, 80 bytes): Xq, cw2, map, sbuf, Xq row pitch in bytes (16 * padded samples; zq rows are
1/128 of it), ldn_bytes, B, nloc, quads, pitch_bytes, n0, -, live record (or null; word 0 = live columns), zq.
"""
import os

CT = 32                                   # samples per staged (c, w) block
SLOT_XOR = 0x4000                         # (c, w) slots of 16 KB at 0x0000 / 0x4000
LDS_BYTES = 0x8000
WG = 512

S_KARG = "s[0:1]"
S_WGX, S_WGY = "s2", "s3"                 # -> node group, column block
S_XP, S_CP, S_MAP, S_SBUF = (4, 5), (6, 7), (8, 9), (10, 11)
S_LDX, S_LDN, S_B, S_NLOC, S_NQ, S_PITCH, S_N0 = "s12", "s13", "s14", "s15", "s16", "s17", "s18"
S_CNT, S_TAIL, S_TMP, S_TMP2, S_CSTEP, S_Q, S_DEAD = "s19", "s20", "s21", "s22", "s23", "s24", "s25"
S_ZP = (26, 27)
S_Z, S_ZN = "s28", "s29"
S_BIG = "s[30:31]"
S_EXEC = "s[32:33]"
S_REC = (38, 39)
XSET = (40, 56)                           # two sets of 16 SGPRs: 4 samples x 4 values
V_TID = 0
V_M, V_S, V_D, V_T, V_U = 2, 6, 10, 14, 18
V_RING = 22                               # 2 sets x 2 pairs x {c, w, c, w}
V_G = 38                                  # staging: two 16-byte pieces
V_CR, V_CW, V_OC, V_OC2 = 46, 47, 48, 49
V_K = 50                                  # Median: both halves -2^24
V_A = V_D
NVGPR = 52


def vp(base, p):
    return f"v[{base + 2 * p}:{base + 2 * p + 1}]"


def sp(base, p):
    return f"s[{base + 2 * p}:{base + 2 * p + 1}]"


def compute(o, mode, xs, cwb):
    """one sample: x in s[xs:xs+3], {c, w} in v[cwb:cwb+1] (Som.cpp:861-867, Transformation.cpp:12,50)"""
    cw = f"v[{cwb}:{cwb + 1}]"
    P = (0, 1)
    if mode == "med":
        # StandardMedianEstimator, 7 operations per pair (vsom_update.hip, VSOM_MED_STEP): the transposed chunk holds
        # x * 2^24 for a Median context (vsom_xq.hip), and t = fma(M, -2^24, x * 2^24) has exactly the sign of x - M
        # (the scaling is exact, the fused difference rounds once and never to zero; NaN stays NaN);
        # p = clamp(t * 2^127) = [t > 0], n = clamp(-t * 2^127) = [t < 0] (DX10_CLAMP off: NaN passes); the four
        # accumulations are exact-product FMAs that round where the reference's separate multiply and add round.
        for p in P:   # t = M * (-2^24) + x * 2^24
            o.append(f"\tv_pk_fma_f32 {vp(V_D, p)}, {vp(V_M, p)}, v[{V_K}:{V_K + 1}], {sp(xs, p)}")
        for p in P:   # p = [t > 0]
            o.append(f"\tv_pk_mul_f32 {vp(V_T, p)}, {vp(V_D, p)}, {S_BIG} clamp")
        for p in P:   # n = [t < 0]
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
    for p in P:   # delta = x - M
        o.append(f"\tv_pk_add_f32 {vp(V_D, p)}, {sp(xs, p)}, {vp(V_M, p)} neg_lo:[0,1] neg_hi:[0,1]")
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


def compute_zero(o, mode, cwb):
    """the step of a sample whose four values are all +-0: delta = -M, signs cancel exactly in every product
    (gen_update_asm.py, compute_zero_x)"""
    cw = f"v[{cwb}:{cwb + 1}]"
    P = (0, 1)
    for p in P:       # u = w*M (M before the step)
        o.append(f"\tv_pk_mul_f32 {vp(V_U, p)}, {cw}, {vp(V_M, p)} op_sel:[1,0]")
    if mode == "std":
        for p in P:   # t = c*M
            o.append(f"\tv_pk_mul_f32 {vp(V_T, p)}, {cw}, {vp(V_M, p)} op_sel_hi:[0,1]")
        for p in P:   # u = u*M
            o.append(f"\tv_pk_mul_f32 {vp(V_U, p)}, {vp(V_U, p)}, {vp(V_M, p)}")
        for p in P:   # M = M - t
            o.append(f"\tv_pk_add_f32 {vp(V_M, p)}, {vp(V_M, p)}, {vp(V_T, p)} neg_lo:[0,1] neg_hi:[0,1]")
        for p in P:   # S = S + u
            o.append(f"\tv_pk_add_f32 {vp(V_S, p)}, {vp(V_S, p)}, {vp(V_U, p)}")
        return
    for p in P:       # S = u*M + S
        o.append(f"\tv_pk_fma_f32 {vp(V_S, p)}, {vp(V_U, p)}, {vp(V_M, p)}, {vp(V_S, p)}")
    if mode == "sfma":
        for p in P:   # t = c*M ; M = M - t
            o.append(f"\tv_pk_mul_f32 {vp(V_T, p)}, {cw}, {vp(V_M, p)} op_sel_hi:[0,1]")
        for p in P:
            o.append(f"\tv_pk_add_f32 {vp(V_M, p)}, {vp(V_M, p)}, {vp(V_T, p)} neg_lo:[0,1] neg_hi:[0,1]")
    else:
        for p in P:   # M = (-c)*M + M
            o.append(f"\tv_pk_fma_f32 {vp(V_M, p)}, {cw}, {vp(V_M, p)}, {vp(V_M, p)} op_sel_hi:[0,1,1] neg_lo:[1,0,0] neg_hi:[1,0,0]")


_lab = [0]


def step(o, mode, xs, cwb, bit):
    """one sample, the zero form when bit `bit` of the block's mask word is set"""
    if mode == "med" or os.environ.get("VSOM_GEN_NT_NOZ"):
        return compute(o, mode, xs, cwb)
    _lab[0] += 1
    n = _lab[0]
    o.append(f"\ts_bitcmp1_b32 {S_Z}, {bit}")
    o.append(f"\ts_cbranch_scc1 .Lz{n}")
    compute(o, mode, xs, cwb)
    o.append(f"\ts_branch .Le{n}")
    o.append(f".Lz{n}:")
    compute_zero(o, mode, cwb)
    o.append(f".Le{n}:")


def kernel(name, mode):
    o = []
    E = o.append
    E(f"\t.text\n\t.globl {name}\n\t.p2align 8\n\t.type {name},@function\n{name}:")
    E(f"\ts_load_dwordx8 s[4:11], {S_KARG}, 0x0")              # Xq, cw2, map, sbuf
    E(f"\ts_load_dwordx4 s[12:15], {S_KARG}, 0x20")            # xq row pitch, ldn_bytes, B, nloc
    E(f"\ts_load_dwordx2 s[16:17], {S_KARG}, 0x30")            # quads, pitch_bytes
    E(f"\ts_load_dword {S_N0}, {S_KARG}, 0x38")
    E(f"\ts_load_dwordx2 s[{S_REC[0]}:{S_REC[1]}], {S_KARG}, 0x40")
    E(f"\ts_load_dwordx2 s[{S_ZP[0]}:{S_ZP[1]}], {S_KARG}, 0x48")
    E(f"\ts_and_b32 {S_TMP}, {S_WGX}, 7")                      # XCD label
    E(f"\ts_lshr_b32 {S_TMP2}, {S_WGX}, 3")                    # column block
    E(f"\ts_lshl_b32 {S_WGY}, {S_WGY}, 3")
    E(f"\ts_add_u32 {S_WGX}, {S_WGY}, {S_TMP}")                # node group = wgy*8 + xcd
    E(f"\ts_mov_b32 {S_WGY}, {S_TMP2}")
    E(f"\tv_and_b32_e32 v{V_TID}, 0x3ff, v{V_TID}")
    E(f"\tv_readfirstlane_b32 {S_Q}, v{V_TID}")
    E(f"\ts_lshr_b32 {S_Q}, {S_Q}, 6")                         # wavefront of the workgroup
    E(f"\ts_lshl_b32 {S_TMP}, {S_WGY}, 3")
    E(f"\ts_add_u32 {S_Q}, {S_Q}, {S_TMP}")                    # this wavefront's column quad
    E(f"\ts_waitcnt lgkmcnt(0)")
    E(f"\ts_cmp_eq_u64 s[{S_REC[0]}:{S_REC[1]}], 0")
    E(f"\ts_cbranch_scc1 .L_nq_{name}")
    E(f"\ts_load_dword {S_NQ}, s[{S_REC[0]}:{S_REC[1]}], 0x0")   # live columns (device value)
    E(f"\ts_waitcnt lgkmcnt(0)")
    E(f"\ts_add_u32 {S_NQ}, {S_NQ}, 3")
    E(f"\ts_lshr_b32 {S_NQ}, {S_NQ}, 2")
    E(f".L_nq_{name}:")
    E(f"\ts_lshl_b32 {S_TMP}, {S_WGY}, 3")                     # the whole column block beyond the quads: leave
    E(f"\ts_cmp_ge_u32 {S_TMP}, {S_NQ}")
    E(f"\ts_cbranch_scc1 .L_end_{name}")
    E(f"\ts_lshl_b32 {S_TMP}, {S_WGX}, 6")                     # first node of the workgroup
    E(f"\ts_cmp_ge_u32 {S_TMP}, {S_NLOC}")
    E(f"\ts_cbranch_scc1 .L_end_{name}")
    E(f"\ts_cmp_ge_u32 {S_Q}, {S_NQ}")                         # a dead quad inside a live block: takes part in the
    E(f"\ts_cselect_b32 {S_DEAD}, 1, 0")                       # staging and the barriers, stores nothing
    if mode == "med":
        E(f"\ts_mov_b32 s30, 0x7f000000")                         # 2^127
        E(f"\ts_mov_b32 s31, 0x7f000000")
        E(f"\tv_mov_b32_e32 v{V_K}, 0xcb800000")                  # -2^24
        E(f"\tv_mov_b32_e32 v{V_K + 1}, 0xcb800000")
    # x / mask rows of this quad
    E(f"\ts_mul_i32 {S_TMP}, {S_Q}, {S_LDX}")
    E(f"\ts_mul_hi_u32 {S_TMP2}, {S_Q}, {S_LDX}")
    E(f"\ts_add_u32 s{S_XP[0]}, s{S_XP[0]}, {S_TMP}")
    E(f"\ts_addc_u32 s{S_XP[1]}, s{S_XP[1]}, {S_TMP2}")
    E(f"\ts_lshr_b32 {S_TMP}, {S_LDX}, 7")
    E(f"\ts_mul_i32 {S_TMP}, {S_TMP}, {S_Q}")
    E(f"\ts_add_u32 s{S_ZP[0]}, s{S_ZP[0]}, {S_TMP}")
    E(f"\ts_addc_u32 s{S_ZP[1]}, s{S_ZP[1]}, 0")
    # (c, w): staging piece of thread t = pair-row t>>6 (+8), node t&63
    E(f"\tv_and_b32_e32 v{V_CR}, 63, v{V_TID}")
    E(f"\tv_lshlrev_b32_e32 v{V_CR}, 4, v{V_CR}")              # lane*16: read base (slot 0)
    E(f"\tv_lshlrev_b32_e32 v{V_CW}, 4, v{V_TID}")             # write: tid*16 (+8192)
    E(f"\tv_lshrrev_b32_e32 v{V_OC}, 6, v{V_TID}")
    E(f"\tv_mul_lo_u32 v{V_OC}, v{V_OC}, {S_LDN}")
    E(f"\tv_add_u32_e32 v{V_OC}, v{V_OC}, v{V_CR}")
    E(f"\ts_lshl_b32 {S_TMP}, {S_LDN}, 3")
    E(f"\tv_add_u32_e32 v{V_OC2}, {S_TMP}, v{V_OC}")
    E(f"\ts_lshl_b32 {S_TMP}, {S_WGX}, 10")                    # node group * 64 nodes * 16 B
    if os.environ.get("VSOM_GEN_NT_CWL2"):                      # development, timing only (WRONG results): every workgroup
        E(f"\ts_and_b32 {S_TMP}, {S_TMP}, 0x400")                # stages node group 0 / 1's (c, w): 8 MB, L2-resident
    E(f"\ts_add_u32 s{S_CP[0]}, s{S_CP[0]}, {S_TMP}")
    E(f"\ts_addc_u32 s{S_CP[1]}, s{S_CP[1]}, 0")
    E(f"\ts_lshl_b32 {S_CSTEP}, {S_LDN}, 4")                   # 16 pair-rows
    for r in range(V_M, V_M + 8):                               # currentModel / currentModelSigma .setZero() :843-844
        E(f"\tv_mov_b32_e32 v{r}, 0")
    E(f"\ts_cmp_eq_u32 {S_B}, 0")
    E(f"\ts_cbranch_scc1 .L_store_{name}")
    E(f"\ts_add_u32 {S_CNT}, {S_B}, {CT - 1}")
    E(f"\ts_lshr_b32 {S_CNT}, {S_CNT}, 5")
    E(f"\ts_sub_u32 {S_CNT}, {S_CNT}, 1")                      # full blocks before the last one
    E(f"\ts_lshl_b32 {S_TMP}, {S_CNT}, 5")
    E(f"\ts_sub_u32 {S_TAIL}, {S_B}, {S_TMP}")                 # samples of the last block: 1..32

    def gload():
        E(f"\tglobal_load_dwordx4 v[{V_G}:{V_G + 3}], v{V_OC}, s[{S_CP[0]}:{S_CP[1]}]")
        E(f"\tglobal_load_dwordx4 v[{V_G + 4}:{V_G + 7}], v{V_OC2}, s[{S_CP[0]}:{S_CP[1]}]")
        E(f"\ts_add_u32 s{S_CP[0]}, s{S_CP[0]}, {S_CSTEP}")
        E(f"\ts_addc_u32 s{S_CP[1]}, s{S_CP[1]}, 0")

    def lwrite():
        E(f"\tds_write_b128 v{V_CW}, v[{V_G}:{V_G + 3}]")
        E(f"\tds_write_b128 v{V_CW}, v[{V_G + 4}:{V_G + 7}] offset:8192")

    def xload(st):
        E(f"\ts_load_dwordx16 s[{st}:{st + 15}], s[{S_XP[0]}:{S_XP[1]}], 0x0")
        E(f"\ts_add_u32 s{S_XP[0]}, s{S_XP[0]}, 64")
        E(f"\ts_addc_u32 s{S_XP[1]}, s{S_XP[1]}, 0")

    def cread(ring, g):
        """{c, w} of the two sample pairs of group g -> ring set `ring`"""
        r = V_RING + 8 * ring
        E(f"\tds_read_b128 v[{r}:{r + 3}], v{V_CR} offset:{1024 * (2 * g)}")
        E(f"\tds_read_b128 v[{r + 4}:{r + 7}], v{V_CR} offset:{1024 * (2 * g + 1)}")

    # ---- prologue: (c, w) block 0 -> slot 0, block 1 -> registers; x of group 0; mask word 0 ------
    gload()
    xload(XSET[0])
    E(f"\ts_load_dword {S_Z}, s[{S_ZP[0]}:{S_ZP[1]}], 0x0")
    E(f"\ts_add_u32 s{S_ZP[0]}, s{S_ZP[0]}, 4")
    E(f"\ts_addc_u32 s{S_ZP[1]}, s{S_ZP[1]}, 0")
    E(f"\ts_mov_b32 {S_ZN}, 0")
    E(f"\ts_waitcnt vmcnt(0)")
    lwrite()
    E(f"\tv_xor_b32_e32 v{V_CW}, {SLOT_XOR}, v{V_CW}")
    E(f"\ts_waitcnt lgkmcnt(0)")
    E(f"\ts_cmp_eq_u32 {S_CNT}, 0")
    E(f"\ts_cbranch_scc1 .L_p1_{name}")
    gload()
    E(f".L_p1_{name}:")
    E(f"\ts_barrier")
    cread(0, 0)
    E(f"\ts_cmp_eq_u32 {S_CNT}, 0")
    E(f"\ts_cbranch_scc1 .L_last_{name}")
    # ---- main loop: one full block of 32 samples per iteration -----------------------------------
    E(f"\t.p2align 6\n.L_loop_{name}:")
    E(f"\ts_waitcnt vmcnt(0)")                                # block b+1 landed in the staging registers
    lwrite()                                                    # -> the slot block b-1 was read from
    # a dead quad (784 dims: the last workgroup of every node group has four) only stages: its chain steps were 2 % of
    # the launch's issue slots
    E(f"\ts_cmp_lg_u32 {S_DEAD}, 0")
    E(f"\ts_cbranch_scc1 .L_dead_{name}")
    for g in range(CT // 4):
        E(f"\ts_waitcnt lgkmcnt(0)")                          # x and (c, w) of this group (and my LDS writes)
        if g == 0:
            E(f"\ts_cmp_lt_u32 {S_CNT}, 2")                   # block b+2 -> registers (if there is one)
            E(f"\ts_cbranch_scc1 .L_nl_{name}")
            gload()
            E(f".L_nl_{name}:")
            E(f"\ts_load_dword {S_ZN}, s[{S_ZP[0]}:{S_ZP[1]}], 0x0")   # next block's mask word
            E(f"\ts_add_u32 s{S_ZP[0]}, s{S_ZP[0]}, 4")
            E(f"\ts_addc_u32 s{S_ZP[1]}, s{S_ZP[1]}, 0")
        xload(XSET[(g + 1) % 2])                                # next group's x (the next block's at g = 7)
        if g + 1 < CT // 4:
            cread((g + 1) % 2, g + 1)
        xs, r = XSET[g % 2], V_RING + 8 * (g % 2)
        for i in range(4):
            step(o, mode, xs + 4 * i, r + 2 * i, 4 * g + i)
    E(f"\tv_xor_b32_e32 v{V_CW}, {SLOT_XOR}, v{V_CW}")
    E(f"\tv_xor_b32_e32 v{V_CR}, {SLOT_XOR}, v{V_CR}")
    E(f"\ts_waitcnt lgkmcnt(0)")
    E(f"\ts_mov_b32 {S_Z}, {S_ZN}")
    E(f"\ts_barrier")                                          # slot b+1 written by all, slot b read by all
    cread(0, 0)
    E(f"\ts_sub_u32 {S_CNT}, {S_CNT}, 1")
    E(f"\ts_cmp_lg_u32 {S_CNT}, 0")
    E(f"\ts_cbranch_scc1 .L_loop_{name}")
    E(f"\ts_branch .L_last_{name}")
    # ---- the same block for a dead quad: staging, the barrier, nothing else ------------------------
    E(f".L_dead_{name}:")
    E(f"\ts_cmp_lt_u32 {S_CNT}, 2")                           # block b+2 -> registers (if there is one)
    E(f"\ts_cbranch_scc1 .L_dnl_{name}")
    gload()
    E(f".L_dnl_{name}:")
    E(f"\tv_xor_b32_e32 v{V_CW}, {SLOT_XOR}, v{V_CW}")
    E(f"\ts_waitcnt lgkmcnt(0)")
    E(f"\ts_barrier")
    E(f"\ts_sub_u32 {S_CNT}, {S_CNT}, 1")
    E(f"\ts_cmp_lg_u32 {S_CNT}, 0")
    E(f"\ts_cbranch_scc1 .L_loop_{name}")
    E(f"\ts_branch .L_end_{name}")
    # ---- last block: 1..32 samples, no staging ----------------------------------------------------
    E(f".L_last_{name}:")
    E(f"\ts_cmp_lg_u32 {S_DEAD}, 0")                           # (a dead quad of a chunk of at most 32 samples)
    E(f"\ts_cbranch_scc1 .L_end_{name}")
    for g in range(CT // 4):
        E(f"\ts_cmp_le_u32 {S_TAIL}, {4 * g}")
        E(f"\ts_cbranch_scc1 .L_store_{name}")
        E(f"\ts_waitcnt lgkmcnt(0)")
        if g + 1 < CT // 4:
            xload(XSET[(g + 1) % 2])                            # (rows past the chunk: padded, never consumed)
            cread((g + 1) % 2, g + 1)
        xs, r = XSET[g % 2], V_RING + 8 * (g % 2)
        for i in range(4):
            if i > 0:
                E(f"\ts_cmp_le_u32 {S_TAIL}, {4 * g + i}")
                E(f"\ts_cbranch_scc1 .L_store_{name}")
            step(o, mode, xs + 4 * i, r + 2 * i, 4 * g + i)
    # ---- epilogue: map row <- M (Som.cpp:870), sigma buffer <- raw S --------------------------------
    E(f".L_store_{name}:")
    E(f"\ts_waitcnt vmcnt(0) lgkmcnt(0)")
    E(f"\ts_cmp_lg_u32 {S_DEAD}, 0")
    E(f"\ts_cbranch_scc1 .L_end_{name}")
    VN = f"v{V_A + 2}"
    VA = f"v[{V_A}:{V_A + 1}]"
    E(f"\tv_and_b32_e32 {VN}, 63, v{V_TID}")
    E(f"\ts_lshl_b32 {S_TMP}, {S_WGX}, 6")
    E(f"\tv_add_u32_e32 {VN}, {S_TMP}, {VN}")                   # local node index
    E(f"\tv_cmp_gt_u32_e32 vcc, {S_NLOC}, {VN}")
    E(f"\ts_and_saveexec_b64 {S_EXEC}, vcc")
    E(f"\ts_cbranch_execz .L_end_{name}")
    E(f"\tv_add_u32_e32 {VN}, {S_N0}, {VN}")                    # global node index
    E(f"\ts_lshl_b32 {S_TMP}, {S_Q}, 4")                       # quad * 16 B
    for base, tag in ((S_MAP, V_M), (S_SBUF, V_S)):
        E(f"\ts_add_u32 {S_TMP2}, s{base[0]}, {S_TMP}")
        E(f"\ts_addc_u32 s34, s{base[1]}, 0")
        E(f"\tv_mov_b32_e32 v{V_A}, {S_TMP2}")
        E(f"\tv_mov_b32_e32 v{V_A + 1}, s34")
        E(f"\tv_mad_u64_u32 {VA}, s[34:35], {VN}, {S_PITCH}, {VA}")
        E(f"\tglobal_store_dwordx4 {VA}, v[{tag}:{tag + 3}], off")
    E(f".L_end_{name}:")
    E(f"\ts_endpgm")
    E(f".L_func_end_{name}:")
    E(f"\t.size {name}, .L_func_end_{name}-{name}")
    return "\n".join(o)


MODES = ("std", "fma", "sfma", "med")


def emit():
    """[(name, text, vgprs, kernarg bytes, lds bytes, dx10_clamp, workgroup size)] for gen_update_asm.main()"""
    out = []
    for m in MODES:
        name = f"vsom_update_{m}_nt4_gfx950"
        out.append((name, kernel(name, m), NVGPR, 80, LDS_BYTES, 0 if m == "med" else 1, WG))
    return out
