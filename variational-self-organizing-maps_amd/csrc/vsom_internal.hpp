// vsom_internal.hpp -- shared declarations of libvsom_hip.so (gfx950 only).
//
// Layout in HBM (see DESIGN.md "Data layout"):
//   model rows   : N x pitch fp32, pitch = nparts * part_pitch, part_pitch = roundup(part_len,32)
//                  Standard/Median: one part of D floats; CLR: [A(P) | B(P)], P = D/2.
//                  pad columns are kept at zero.
//   samples      : Bcap x xpitch fp32 (xpitch = roundup(J,32), zero padded)
//   CLR samples  : XP/YP Bcap x part_pitch (x'_p = x[i(p)], y'_p = x[j(p)], pairs i<j
//                  lexicographic, Transformation.cpp:94-101)
//   cw           : (c = w/W prefix, w) per (sample, node), pair-interleaved: one float4
//                  {c_j, w_j, c_j+1, w_j+1} at [(j>>1)][node], ceil(B/2)+8 pair rows of ldn nodes
//   Xnext[2]     : raw B x J chunks landing on the copy stream (double-buffered ingest)
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>

#include "../../include/vsom_hip.h"

typedef unsigned long long u64;

// lane = node update kernels yield ceil(N/64)*ceil(D/14) wavefronts; at or below this many the
// (node, dim pair)-per-lane chain kernels are used instead (1024 SIMDs on the chip).  While every wavefront
// has a SIMD to itself the assembly kernels take ~0.12 us per sample whatever the map, the LDS-staged chain
// kernel ~2.2e-13 s per (node, dim, sample): they meet near N*D = 4e5 (tools/exp/chain_crossover.py:
// 4096 x 128: 0.99 vs 1.12 ms at B = 8192; 4096 x 64: ~2 vs 0.75 ms at B = 16384)
#define VSOM_CHAIN_MAX_WAVES 448

#define VSOM_TK 32          // K-chunk of the tile kernels; row pitches are multiples of it

// spare rows behind the staged sample matrices (readable, contents unspecified: zero at allocation, stale
// samples after a larger chunk): the pipelined update kernels read ahead into them and never consume them
constexpr size_t VSOM_ROW_PAD = 32;
// bytes of vsom_ctx::onl_state (layout: vsom_online.hip)
constexpr size_t VSOM_ONL_STATE_BYTES = 4352;

struct vsom_ctx {
    int device = 0;
    uint32_t W = 0, H = 0, J = 0, D = 0, N = 0;
    int transform = 0;
    uint32_t nparts = 1, part_len = 0, part_pitch = 0, pitch = 0;
    uint32_t xpitch = 0;
    int bmu_mode = VSOM_BMU_AUTO;

    hipStream_t own_stream = nullptr, stream = nullptr;
    // side stream for work that only has to finish before the next entry point (the MSE sum)
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool aux_pending = false;

    // model state
    float *map = nullptr, *sigma = nullptr, *S = nullptr, *weight = nullptr;
    u64 *hits = nullptr;

    // chunk
    size_t B = 0, Bcap = 0;
    bool chunk_loaded = false;      // a chunk (possibly of 0 rows) has been staged
    float *Xs = nullptr, *XP = nullptr, *YP = nullptr;
    float *Xraw = nullptr;          // staging for host uploads (B x J, unpadded)
    size_t Xraw_cap = 0;
    // double-buffered ingest: raw chunks land here on copy_stream while the current chunk trains
    float *Xnext[2] = {nullptr, nullptr};
    size_t Xnext_cap[2] = {0, 0};
    size_t Bnext = 0;
    int next_slot = 0;              // slot the next prefetch writes
    int ready_slot = -1;            // slot holding a prefetched, not yet committed chunk
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_copied[2] = {nullptr, nullptr};    // H2D copy into slot finished
    hipEvent_t ev_staged[2] = {nullptr, nullptr};    // staging kernels that read slot finished
    bool staged_valid[2] = {false, false};
    u64 *lastbmu = nullptr;
    float *sqres = nullptr;
    // The NEXT chunk staged beside the epoch of the current one (vsom_prefetch_chunk / vsom_stage_next_device -> copy
    // stream; vsom_commit_chunk adopts it).  Once a phase 2 has built its transposed chunk nothing of the epoch reads the
    // staged rows (Xs, Xc, the int8 images) any more -- `ev_rows_free` -- so the staging kernels of chunk k+1 may
    // overwrite them while the chains of chunk k run; what phase 2 still reads is double-buffered: lastBMU and the
    // compaction's column record (the *_alt set is what the ahead staging writes, commit swaps the two).
    u64 *lastbmu_alt = nullptr;
    int *cc_idx_alt = nullptr, *cc_inv_alt = nullptr;
    unsigned *cc_meta_alt = nullptr;
    hipEvent_t ev_rows_free = nullptr, ev_ahead = nullptr;
    bool rows_free_valid = false;   // ev_rows_free belongs to the last enqueued work on this context
    bool ahead_valid = false;       // a chunk is staged ahead (ahead_B rows; its compaction / int8-image state below)
    // staging kernels of a chunk staged ahead have been launched over the staged-row buffers and `stream` has neither
    // adopted them nor staged a chunk of its own since (whatever ahead_valid says: the ahead chunk may have been abandoned):
    // `stream` must wait for ev_ahead before it writes those buffers, and nobody may read the current chunk's rows
    bool ahead_rows = false;
    size_t ahead_B = 0;
    bool ahead_cc = false, ahead_xi = false;
    const float *next_dev = nullptr;   // vsom_stage_next_device: rows in HBM waiting for vsom_commit_chunk
    size_t next_dev_B = 0;
    bool next_dev_pending = false;
    float *mse = nullptr;           // [1]  pinned HOST memory (device-visible): kernels store, vsom_get_mse reads after a stream wait
    int *pair_i = nullptr, *pair_j = nullptr;   // CLR pair tables [P]

    // BMU tile-search scratch
    u64 *partial = nullptr; size_t partial_cap = 0;
    unsigned char *nan0 = nullptr;

    // duplicate-row representatives of the exact search (vsom_bmu.hip, bmu_dedupe_*)
    void *dd_hash = nullptr; int *dd_rep = nullptr, *dd_list = nullptr;
    bool dedupe = true;             // VSOM_NO_DEDUPE=1 switches it off (A/B measurements)
    double dd_min_work = 2.0e10;    // exact searches of at least this many (sample, node, value) triples (vsom_set_row_dedupe)
    bool tiny_lds_attr[12] = {};    // online_tiny_chunk_kernel<kind, local, U>: dynamic LDS limit raised (per context: its device, its kind)

    // MFMA shortlist scratch
    float *sl_G = nullptr; size_t sl_cap = 0; float *sl_nrm = nullptr; unsigned *sl_scal = nullptr;
    float *sl_a2 = nullptr;         // CLR shortlist: per-node max A^2 (the select kernel's per-node bounds)
    int *sl_list = nullptr; size_t sl_list_cap = 0;
    float *sl_tmin = nullptr; size_t sl_tmin_cap = 0;
    unsigned *sl_fb = nullptr;      // pinned host feedback: {redo samples, candidates, rows, seq}
    float *sl_fs = nullptr, *sl_fm = nullptr;      // CLR shortlist: sample / node feature rows (vsom_shortlist.hip)
    size_t sl_fs_cap = 0, sl_fm_cap = 0;           // bytes
    // integer contraction of the shortlist (vsom_sl_i8.hip): int8 images of the chunk / the model rows
    signed char *sl_xi = nullptr; size_t sl_xi_cap = 0; float *sl_l1 = nullptr;
    signed char *sl_q = nullptr; double *sl_qscale = nullptr, *sl_qcorr = nullptr;
    void *sl_qfast = nullptr;       // int4 per node: the constants of the uint8 kind's fp32 epilogue (sl_i8_value_fast)
    uint32_t sl_kp8 = 0;
    bool xi_valid = false;          // sl_xi / sl_l1 describe the staged chunk
    int sl_par = 0;                 // which of the two scal sets the next search uses
    int sl_skip = 0;
    unsigned sl_seq_seen = 0;       // feedback sequence number already acted on
    int sl_fail_streak = 0;         // consecutive probes that had to redo most samples exactly

    // neighbourhood
    float2 *cw = nullptr; size_t cw_cap = 0;
    float *lut = nullptr; size_t lut_cap = 0; float *lut_host = nullptr;   // lut_host: 2 x lut_cap, pinned
    hipEvent_t lut_ev[2] = {nullptr, nullptr}; bool lut_ev_valid[2] = {false, false}; int lut_slot = 0;
    double lut_sigma = -1.0; uint32_t lut_w = 0, lut_h = 0;
    double *lutd = nullptr; size_t lutd_cap = 0; double lutd_sigma = -1.0;   // online path (double)
    // the table's host image: two pinned slots used alternately (the device copy is enqueued on the stream, one-launch
    // chunks of tiny maps read the slot itself), each guarded by an event recorded behind its last reader
    double *lutd_host = nullptr; size_t lutd_host_cap = 0; double lutd_host_sigma[2] = {-1.0, -1.0};
    hipEvent_t lutd_ev[2] = {nullptr, nullptr}; bool lutd_ev_valid[2] = {false, false}; int lutd_slot = 0;

    // hand-scheduled update kernel (code object loaded with hipModuleLoadData)
    void *upd_module = nullptr, *upd_clr8 = nullptr, *upd_nt[4] = {nullptr, nullptr, nullptr, nullptr};   // nt: std, fma, sfma, med
    int update_mode = VSOM_UPDATE_STRICT;
    bool use_chain = true;
    bool use_tiny = true;           // one-launch epoch for tiny maps (VSOM_NO_TINY=1 disables, debugging)

    // column compaction (vsom_compact.hip): columns that are zero in every row of the chunk are retired exactly
    unsigned *cc_flags = nullptr;   // [xpitch] live flags
    int *cc_idx = nullptr;          // [cpitch] live column list, -1 beyond the live count
    int *cc_inv = nullptr;          // [xpitch] column -> compacted position or -1
    unsigned *cc_meta = nullptr;    // device: {live columns, live 14-dim slices, live columns rounded up to 32, seq}
    unsigned *cc_fb = nullptr;      // pinned host mirror: {live columns, seq}
    unsigned cc_seen = 0;
    int cc_skip = 0;
    long cc_min_rows = 1024;        // chunks with fewer rows are not compacted (< 0: never)
    bool cc_valid = false;          // the staged chunk has a compaction (Xc, cc_idx, cc_meta describe it)
    uint32_t cpitch = 0;            // row pitch of the compacted matrices
    float *Xc = nullptr; size_t Xc_cap = 0;      // (Bcap + VSOM_ROW_PAD) x cpitch
    float *Mc = nullptr;            // N x cpitch: model rows on the live columns (search)
    float *Uc_map = nullptr, *Uc_S = nullptr;    // N x cpitch: the chains' M and raw S on the live columns
    // the chunk transposed into column quads (vsom_xq.hip) for the lane = node, four-dims-per-wavefront chain kernels
    float *Xq = nullptr; size_t Xq_cap = 0;      // [quads rounded up to 8][bpad] float4
    unsigned *zq = nullptr;                      // [quads][bpad / 32] all-zero (sample, quad) bits
    bool xq_valid = false;
    uint32_t xq_bpad = 0, xq_quads = 0;

    // online path scratch
    float *v_dev = nullptr;         // one sample, padded
    float *v_pinned = nullptr;      // its pinned host staging (+ 16 floats for results)
    float *res_dev = nullptr;       // residual
    u64 *onl_state = nullptr;       // argmin key slots + flags of the online scan (vsom_online.hip)
    float *onl_f = nullptr;         // [4]: dist, mse
    // image-bounded search of the online chunk loop (vsom_online.hip): one byte per model value + 4 scalars per node,
    // lower bounds of the sample being searched, min-upper-bound slots, per-sample bound terms
    unsigned char *onl_img = nullptr; void *onl_nsc = nullptr; float *onl_lb = nullptr; unsigned *onl_u = nullptr;
    void *onl_xsc = nullptr; size_t onl_xsc_cap = 0;
    unsigned char *onl_dirty = nullptr;      // [N] nodes whose sigmaMap row is written at the end of the chunk

    // pinned staging of vsom_set_state's host arrays
    void *st_pinned = nullptr; size_t st_pinned_cap = 0;
    void *out_pinned = nullptr;     // [8192] u64: vsom_get_last_bmu of short chunks
    // device scratch of the distance queries (vsom_distances / _row / _raw): grow-only
    void *q_scratch = nullptr; size_t q_scratch_cap = 0;

    // timing
    uint32_t timing = 0;            // bit (1u << VSOM_T_*): that kernel group is timed with HIP events
    struct Ev { hipEvent_t a, b; int which; };
    std::vector<Ev> ev_live;
    std::vector<Ev> ev_pool;
    float t_ms[VSOM_T_COUNT] = {0};
    uint32_t t_cnt[VSOM_T_COUNT] = {0};
};

// error plumbing -----------------------------------------------------------------------------
void vsom_set_error(const std::string &msg);
int vsom_fail(int code, const std::string &msg);
// (an allocation that does not fit is VSOM_ERR_NOMEM and leaves the context usable: every allocation site drops the old
// buffer and its capacity first, and the runtime's last-error slot is cleared so that the next hipGetLastError() of a
// launch sequence does not report it again)
#define VSOM_HIP_CHECK(expr)                                                             \
    do {                                                                                 \
        hipError_t _e = (expr);                                                          \
        if (_e != hipSuccess) {                                                          \
            (void)hipGetLastError();                                                     \
            return vsom_fail(_e == hipErrorOutOfMemory ? VSOM_ERR_NOMEM : VSOM_ERR_HIP,  \
                             std::string(#expr) + ": " + hipGetErrorString(_e));         \
        }                                                                                \
    } while (0)

struct TimerScope {
    vsom_ctx *c; int which; vsom_ctx::Ev ev; bool on;
    TimerScope(vsom_ctx *ctx, int w);
    ~TimerScope();
};

// kernel launchers (each enqueues on ctx->stream and returns a vsom_status) -------------------
int launch_stage_chunk(vsom_ctx *c, const float *x_dev, size_t B);
// the same kernels for the NEXT chunk on the copy stream, beside the running epoch (false: not possible now --
// the caller stages at commit time instead)
bool vsom_can_stage_ahead(const vsom_ctx *c, size_t B);
int launch_stage_chunk_ahead(vsom_ctx *c, const float *x_dev, size_t B);
int vsom_adopt_ahead(vsom_ctx *c);
// pieces of the double-buffered ingest shared with the multi-GPU group (vsom_capi.hip)
int vsom_prefetch_rows(vsom_ctx *c, const float *x_host, size_t B, size_t r0, size_t r1);
int vsom_commit_begin(vsom_ctx *c, float **raw, size_t *B);
int vsom_commit_end(vsom_ctx *c);
int launch_bmu_full(vsom_ctx *c, size_t s0, size_t s1);           // findBmu for samples [s0,s1)
int launch_bmu_local(vsom_ctx *c, size_t s0, size_t s1);          // findLocalBmu
int launch_pair_dist(vsom_ctx *c, const u64 *nodes_dev, const u64 *rows_dev, size_t count,
                     float *out_dev);
int launch_finish(vsom_ctx *c);
int vsom_join_aux(vsom_ctx *c);      // make ctx->stream wait for the side stream's pending work
int launch_bmu_restricted(vsom_ctx *c, u64 min_hits);
int launch_row_dist(vsom_ctx *c, size_t row, float *out_dev);
int launch_raw_dist(vsom_ctx *c, const u64 *nodes_dev, const u64 *vrows_dev, size_t count, int from_map,
                    float *out_dev);
int launch_phase2(vsom_ctx *c, double sigma, size_t n0, size_t n1);
int ensure_lut(vsom_ctx *c, double sigma);
// column compaction (vsom_compact.hip)
bool vsom_cc_applies(const vsom_ctx *c);
int vsom_cc_begin(vsom_ctx *c, size_t B, bool *on);
// live-column record + gathered rows (+ int8 images) of a chunk of B rows, on `stream`, into the given record
int vsom_cc_stage(vsom_ctx *c, size_t B, hipStream_t stream, int *idx, int *inv, unsigned *meta, bool *xi_out);
int vsom_cc_gather_map(vsom_ctx *c);
int vsom_cc_ensure_update_scratch(vsom_ctx *c);
int vsom_cc_expand(vsom_ctx *c, size_t n0, size_t nloc);
// vsom_sl_i8.hip: rows onto the live columns (+ int8 images)
int launch_sl_gather_quant(vsom_ctx *c, size_t B, hipStream_t stream, const int *idx, bool *xi_out);
int vsom_xq_ensure(vsom_ctx *c);                                 // vsom_xq.hip
bool vsom_tiny_applies(const vsom_ctx *c);                        // vsom_tiny.hip
int launch_tiny_epoch(vsom_ctx *c, double sigma, int is_first);   // whole batch epoch, one workgroup
