/*
 * vsom_hip.h -- C ABI of libvsom_hip.so: the MI355X (gfx950) implementation of the VSOM
 * training hot path (BMU search + Gaussian-neighbourhood mean / sigma^2 update).
 *
 * The reference (PereUbu7/Variational-Self-Organizing-Maps) has no FFI layer: callers link
 * libsom and use `class Som` (include/SOM.hpp:39-189).  This header is the boundary a
 * maintainer binds instead of src/Som.cpp's CPU loops; every entry point names the reference
 * member function it replaces.  Host-side `Som` / `Transformation` mirrors that call these
 * entry points live in variational-self-organizing-maps_amd/host/ (C++) and
 * variational-self-organizing-maps_amd/som.py (ctypes).  INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++ or torch types.
 *  - every function returns 0 on success, a negative vsom_status otherwise; the message is
 *    available from vsom_last_error() (thread-local).  Nothing throws across the ABI.
 *  - model state is row-major N x D fp32 (N = width*height, node index = y*width + x,
 *    D = Transformation::Length(J)); samples are row-major B x J fp32.
 *  - "host" pointers are ordinary host memory, copied synchronously; "dev" pointers are
 *    device memory on the context's GPU.
 *  - all work is enqueued on the context's HIP stream (vsom_set_stream adopts an external
 *    one, e.g. torch's current stream); entry points that return values to the host
 *    synchronise that stream, the *_async ones do not.
 *  - there is no CPU fallback: if no gfx950 device / code object is usable, vsom_create fails.
 */
#ifndef VSOM_HIP_H
#define VSOM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vsom_ctx vsom_ctx;

typedef enum vsom_status {
    VSOM_OK = 0,
    VSOM_ERR_INVALID = -1, /* bad argument / state (e.g. no chunk loaded)          */
    VSOM_ERR_HIP = -2,     /* HIP runtime error, text in vsom_last_error()         */
    VSOM_ERR_NOMEM = -3,
    VSOM_ERR_UNSUPPORTED = -4
} vsom_status;

/* Transformation factories: src/Transformation.cpp:3-39 / 41-77 / 79-167 */
typedef enum vsom_transform {
    VSOM_STANDARD = 0,
    VSOM_MEDIAN = 1,
    VSOM_CLR = 2
} vsom_transform;

/* Som::WeigthDecayFunction: include/SOM.hpp:70-75 */
typedef enum vsom_decay {
    VSOM_EXPONENTIAL = 0,
    VSOM_INVERSE_PROPORTIONAL = 1,
    VSOM_BATCHMAP = 2
} vsom_decay;

/* BMU search strategy of the full search (results are identical) */
typedef enum vsom_bmu_mode {
    VSOM_BMU_AUTO = 0,      /* MFMA shortlist + exact-order refinement when applicable */
    VSOM_BMU_EXACT = 1,     /* brute-force exact-order VALU kernel                     */
    VSOM_BMU_SHORTLIST = 2  /* force the MFMA shortlist path                           */
} vsom_bmu_mode;

/* arithmetic of the phase-2 chain kernel */
typedef enum vsom_update_mode {
    VSOM_UPDATE_STRICT = 0, /* one rounding per fp32 operation: bit-identical to the reference's SSE2
                               build (default)                                                    */
    VSOM_UPDATE_FMA = 1,    /* contracted Standard chains: M = fma(c,d,M), S = fma(w*d,d,S) (1/3 fewer VALU
                               ops).  After ONE epoch from a given map, map / sigmaMap differ from the reference by
                               rounding only: |err| <= 1e-5 * max(|ref|, scale of the chain's operands) (measured
                               1e-7; pure element-wise <= 4e-7 on the MNIST workloads), BMU indices, bmuHits, MSE
                               and weightMap bit-exact.  NOT a schedule-level guarantee: the next search runs on
                               the perturbed map, near-ties flip, and a multi-epoch trainBatchSom leaves the
                               reference's trajectory within the first epochs (measured: profiles/
                               r3_fma_schedule.jsonl -- C3, 2 chunks: 5 of 8192 BMUs differ in epoch 0, 24 % by
                               epoch 9).  Use for single passes / throughput studies only.                  */
    VSOM_UPDATE_FMA_SIGMA = 2 /* only the variance accumulation (Som.cpp:867) contracted: t = c*d, M = M + t (Som.cpp:864)
                               as the reference rounds them, S = fma(w*d, d, S) (1/6 fewer VALU ops).  map is BIT-IDENTICAL, and so
                               are lastBMU, bmuHits, MSE and weightMap of every later epoch of a schedule -- no
                               training step reads sigmaMap (Transformation.cpp:7-8,45-46,82: the built-in
                               Comparers ignore the dispersion); sigmaMap is a sum of non-negative terms.
                               What can be PROVEN for it: both accumulations of B non-negative terms carry a
                               relative error <= (B+1)*2^-24 against the exact sum, so S differs by at most
                               2(B+1)*2^-24 and sigmaMap = sqrt(S/W) by (B+1)*2^-24 relative = 2.4e-4 at B = 4096
                               (if every rounding pointed the same way; ~sqrt(B)*2^-24 = 4e-6 when they do not).
                               What is MEASURED and asserted: within 1e-5 relative, element by element, over 10-14-
                               epoch schedules up to 128x128x784 with chunks of 4096 (worst 9.1e-7;
                               tests/test_gpu_fma_schedule.py).  The 1e-5 figure is therefore empirical.
                               Median and CLR have ONE arithmetic, bit-identical to the reference, in every mode:
                               the Median chains' fused operations are exact, and the CLR recurrence amplifies
                               rounding differences beyond the tolerance.                                    */
} vsom_update_mode;

/* selectors for vsom_device_ptr / vsom_get_timing */
typedef enum vsom_buffer {
    VSOM_BUF_MAP = 0,      /* float   [N][D]                                    */
    VSOM_BUF_SIGMA = 1,    /* float   [N][D]                                    */
    VSOM_BUF_S = 2,        /* float   [N][D]                                    */
    VSOM_BUF_WEIGHT = 3,   /* float   [N]                                       */
    VSOM_BUF_HITS = 4,     /* uint64  [N]                                       */
    VSOM_BUF_LASTBMU = 5,  /* uint64  [chunk capacity]                          */
    VSOM_BUF_SQRES = 6,    /* float   [chunk capacity]  ||Comparer(x,M[bmu])||^2 */
    VSOM_BUF_CHUNK = 7     /* float   [B][J] staged samples                     */
} vsom_buffer;

typedef enum vsom_timer {
    VSOM_T_STAGE = 0,      /* chunk re-layout kernels                           */
    VSOM_T_BMU = 1,        /* full / local BMU search kernels                   */
    VSOM_T_FINISH = 2,     /* bmuHits + MSE                                     */
    VSOM_T_CW = 3,         /* neighbourhood weight chain (w, w/W) kernel        */
    VSOM_T_UPDATE = 4,     /* mean / sigma^2 chain kernel                       */
    VSOM_T_ONLINE = 5,     /* online (trainSingle) kernels                      */
    VSOM_T_SIGMA = 6,      /* sigmaMap = sqrt(S/W) pass after the assembly chain kernel */
    VSOM_T_COUNT = 7
} vsom_timer;

const char *vsom_last_error(void);
/* number of visible HIP devices (0 when none / no driver) */
int vsom_device_count(void);

/* ---- lifetime -------------------------------------------------------------------------
 * Som::Som(width,height,depth,Transformation) + Som::Construct (SOM.hpp:83-87,
 * Som.cpp:11-48): all state zero.  in_len = J (sample length); D = Length(J).          */
int vsom_create(vsom_ctx **out, int device, uint32_t width, uint32_t height,
                uint32_t in_len, int transform);
void vsom_destroy(vsom_ctx *ctx);
/* adopt an external hipStream_t (NULL = back to the context's own stream) */
int vsom_set_stream(vsom_ctx *ctx, void *hip_stream);
int vsom_synchronize(vsom_ctx *ctx);
int vsom_set_bmu_mode(vsom_ctx *ctx, int mode);
int vsom_set_update_mode(vsom_ctx *ctx, int mode);
/* [MI355X build; no counterpart in the reference, whose phase 2 walks every column: Som.cpp:840-875]
 * Exact retirement of the sample columns that are zero in every row of a chunk (csrc/vsom_compact.hip: their
 * chains stay 0 -- or NaN for a node whose first weight is 0/0 -- and they add nothing to the search's
 * contraction; MNIST has ~120 such columns per 4096-image chunk).  Results are bit-identical with it on or off.
 * Chunks of at least min_rows rows use it (default 1024: below that the passes cost more than they save);
 * min_rows < 0 switches it off. */
int vsom_set_column_compaction(vsom_ctx *ctx, long min_rows);
/* [MI355X build] The exact search (small problems, the shortlist's redo list, VSOM_BMU_EXACT) evaluates ONE representative
 * per class of bit-identical model rows (csrc/vsom_bmu.hip: equal rows give equal distances and the reference's strict `<`
 * keeps the lowest index, Som.cpp:293-304) -- batch training on degenerate chunks leaves such maps (an empty chunk: one
 * class).  Results are bit-identical with it on or off.  Searches of at least min_work (sample, node, value) triples use
 * it (default 2e10; behind a shortlist search -- its redo list -- only for the CLR comparer, whose collapsed maps put whole
 * chunks on that list); 0: always; < 0: never. */
int vsom_set_row_dedupe(vsom_ctx *ctx, double min_work);
/* diagnostics of the last MFMA-shortlist search (synchronises): out[0] = samples that had to be
 * redone by the exact-order kernel, out[1] = shortlisted candidates in total, out[2] = samples
 * searched, out[3] = number of shortlist searches so far */
int vsom_get_shortlist_stats(vsom_ctx *ctx, uint32_t *out /*[4]*/);
uint32_t vsom_depth(const vsom_ctx *ctx);   /* D */
uint32_t vsom_nodes(const vsom_ctx *ctx);   /* N */

/* ---- state (Som::map / sigmaMap / SMap / weightMap / bmuHits, SOM.hpp:56-61) -----------
 * NULL pointers are skipped.  Replaces getNeuron/getSigmaNeuron/getWeigthMap/getBmuHits
 * (Som.cpp:164-212) and the element writes of randomInitialize/load (Som.cpp:977-997).    */
int vsom_set_state(vsom_ctx *ctx, const float *map, const float *sigma, const float *S,
                   const float *weight, const uint64_t *bmu_hits);
int vsom_get_state(vsom_ctx *ctx, float *map, float *sigma, float *S, float *weight,
                   uint64_t *bmu_hits);

/* ---- chunk (DataSet::loadNextDataFromStream, DataSet.cpp:118-160) ----------------------
 * Stages B samples and zeroes lastBMU (DataSet.cpp:136-137).                              */
int vsom_upload_chunk(vsom_ctx *ctx, const float *x_host, size_t B);
/* same without the final wait: copy and staging kernels are enqueued on the context's stream and the call returns; x_host
 * must be pinned (vsom_host_alloc; from pageable memory the copy degrades to a blocking one) and stay unchanged until a
 * call that synchronises the context (vsom_get_mse, vsom_get_last_bmu, vsom_train_online_chunk_fetch ...) has returned.
 * For the first chunk of an epoch, when nothing runs that a prefetch on the copy stream could overlap with. */
int vsom_upload_chunk_async(vsom_ctx *ctx, const float *x_host, size_t B);
/* same, samples already resident in HBM (no PCIe copy) */
int vsom_set_chunk_device(vsom_ctx *ctx, const float *x_dev, size_t B);
/* Double-buffered ingest: the reference reloads every chunk from its loader each
 * epoch (Som.cpp:737, DataSet.cpp:118-160), so the host->device copy of chunk i+1 should run while
 * chunk i trains.  vsom_prefetch_chunk copies B x J floats into the context's NEXT raw device buffer
 * on a copy stream and returns at once when x_host is pinned (vsom_host_alloc); the current chunk is
 * untouched.  vsom_commit_chunk makes the prefetched chunk current: the compute stream waits for the
 * copy, stages it and zeroes lastBMU -- the same state vsom_upload_chunk of the same data leaves.
 * vsom_prefetch_wait blocks until the copy has left x_host (then the host buffer may be rewritten). */
int vsom_host_alloc(void **out, size_t bytes);   /* pinned host memory */
int vsom_host_free(void *p);
int vsom_prefetch_chunk(vsom_ctx *ctx, const float *x_host, size_t B);
/* Staging ahead (round 5): when the call directly follows an asynchronous batch epoch on the current chunk
 * (vsom_batch_epoch_async / vsom_batch_phase2_async, Standard / Median on a map large enough for the lane = node
 * chain kernels but small enough that their launch leaves workgroup slots idle -- at most two rounds of the chip's
 * resident workgroups, e.g. 64x64x784; bigger launches lose more to the company than the overlap saves), the
 * prefetched chunk's staging kernels are enqueued too -- on the copy stream, after the point of the
 * epoch from which nothing reads the current chunk's sample rows any more -- so that they run BESIDE the epoch's
 * chains and vsom_commit_chunk launches nothing.  Between such a prefetch and its commit the current chunk's
 * sample rows are gone: lastBMU, the MSE and the model state of the current chunk stay readable, a search or a
 * distance call on it does not (commit first).  In every other situation the staging happens at commit, as before. */
int vsom_prefetch_wait(vsom_ctx *ctx);
int vsom_commit_chunk(vsom_ctx *ctx);
/* the same for a next chunk that ALREADY lives in HBM (x_dev must stay valid until vsom_commit_chunk has been called):
 * staged beside the running epoch when that is possible now, else at vsom_commit_chunk */
int vsom_stage_next_device(vsom_ctx *ctx, const float *x_dev, size_t B);
/* DataSet::getLastBMU (DataSet.cpp:60-69) */
int vsom_get_last_bmu(vsom_ctx *ctx, uint64_t *out_host);
int vsom_set_last_bmu(vsom_ctx *ctx, const uint64_t *in_host);
int vsom_get_sqres(vsom_ctx *ctx, float *out_host);

/* ---- search ----------------------------------------------------------------------------
 * Som::findBmu for every sample of the chunk (Som.cpp:291-309, distance :124-141);
 * writes lastBMU / sqres on the device; optional host copies.                             */
int vsom_bmu_batch(vsom_ctx *ctx, uint64_t *idx_out_host, float *dist_out_host);
/* Som::findBmu(v) for one host vector (Som.cpp:283-309; the perf harness calls it 1000 times,
 * tests/performance/perf_tests.cpp:181-199): one copy, one scan launch, 32 bytes back.  Does not
 * touch the staged chunk.  dist_out = euclidianWeightedDist(bmu, v) (NaN when node 0 is NaN). */
int vsom_find_bmu(vsom_ctx *ctx, const float *v_host, uint64_t *bmu_out, float *dist_out);
/* Som::euclidianWeightedDist(pos, v, ...) (Som.cpp:124-141) and Som::findLocalBmu(v, ..., lastBMU, ...)
 * (Som.cpp:335-454) for ONE host vector (the perf harness calls them a million times each,
 * tests/performance/perf_tests.cpp:200-222, 269-295): one copy, one single-wavefront launch, 16 bytes back,
 * one synchronisation, no allocation.  Neither touches the staged chunk. */
int vsom_dist_single(vsom_ctx *ctx, const float *v_host, uint64_t node, float *dist_out);
int vsom_find_local_bmu(vsom_ctx *ctx, const float *v_host, uint64_t last_bmu, uint64_t *bmu_out, float *dist_out);
/* Som::findRestrictedBmu(v, ..., minBmuHits, ...) (Som.cpp:313-332: node 0 seeds the search whatever its hits) and the
 * distances Som::findRestrictedBmd walks (Som.cpp:457-487: dist_out_host[N] = euclidianWeightedDist(i, v)) for ONE host
 * vector, the same way: one copy, one scan launch, one synchronisation; the staged chunk is not touched. */
int vsom_find_restricted_bmu(vsom_ctx *ctx, const float *v_host, uint64_t min_hits, uint64_t *bmu_out, float *dist_out);
int vsom_distances_single(vsom_ctx *ctx, const float *v_host, float *dist_out_host);
/* Som::findLocalBmu from the current lastBMU of every sample (Som.cpp:335-454) */
int vsom_bmu_local_batch(vsom_ctx *ctx, uint64_t *idx_out_host, float *dist_out_host);
/* Som::euclidianWeightedDist(pos, v, ...) for `count` (node, sample-row) pairs of the chunk */
int vsom_distances(vsom_ctx *ctx, const uint64_t *nodes_host, const uint64_t *rows_host,
                   size_t count, float *dist_out_host);

/* ---- consumers of the search outside the training loop (SURVEY 8f "next" rows) --------------
 * Som::findRestrictedBmu (Som.cpp:313-332) for every sample of the chunk: argmin over node 0 and
 * the nodes with bmuHits >= min_hits.  Overwrites the chunk's lastBMU / sqres like vsom_bmu_batch. */
int vsom_bmu_restricted_batch(vsom_ctx *ctx, uint64_t min_hits, uint64_t *idx_out_host,
                              float *dist_out_host);
/* Som::euclidianWeightedDist of EVERY node for chunk row `row` (input of findRestrictedBmd,
 * Som.cpp:457-487); dist_out_host[N]. */
int vsom_distances_row(vsom_ctx *ctx, size_t row, float *dist_out_host);
/* Som::euclidianWeightedDistRaw(pos, v, ones, ones) (Som.cpp:143-157) for `count` pairs; v is
 * chunk row vrows[i] (from_map = 0) or model vector vrows[i] (from_map = 1, the U-matrix case). */
int vsom_distances_raw(vsom_ctx *ctx, const uint64_t *nodes_host, const uint64_t *vrows_host,
                       size_t count, int from_map, float *dist_out_host);

/* ---- batch epoch: Som::trainBatchSomEpoch (Som.cpp:756-879) ----------------------------
 * phase 1 (:762-806) over samples [s0,s1): BMU (findBmu when is_first else findLocalBmu)
 *   + per-sample ||residual||^2;  finish: bmuHits += 1, fp32 MSE in sample order;
 * phase 2 (:809-876) over nodes [n0,n1): new map, sigmaMap, weightMap rows.
 * vsom_batch_epoch = phase1(0,B) + finish + phase2(0,N); mse_out may be NULL.
 * The split entry points exist for node/sample sharding across GPUs (multi-GPU).*/
int vsom_batch_phase1_async(vsom_ctx *ctx, size_t s0, size_t s1, int is_first);
int vsom_batch_finish_async(vsom_ctx *ctx);
int vsom_batch_phase2_async(vsom_ctx *ctx, double sigma, size_t n0, size_t n1);
int vsom_batch_epoch_async(vsom_ctx *ctx, double sigma, int is_first);
int vsom_batch_epoch(vsom_ctx *ctx, double sigma, int is_first, float *mse_out);
/* MSE of the last finish / online chunk (synchronises) */
int vsom_get_mse(vsom_ctx *ctx, float *mse_out);

/* ---- online path -----------------------------------------------------------------------
 * Som::trainSingle (Som.cpp:885-947) on one host vector; residual_out has
 * vsom_residual_len() floats (may be NULL).  *last_bmu in/out.                            */
uint32_t vsom_residual_len(const vsom_ctx *ctx);
int vsom_train_single(vsom_ctx *ctx, const float *v_host, double eta, double sigma,
                      uint64_t *last_bmu, int decay_fn, float *residual_out,
                      float *dist_out, uint64_t *bmu_out);
/* inner loop of Som::trainBasicSom over the staged chunk (Som.cpp:1159-1171): B sequential
 * trainSingle steps + addBmu (:1189-1192) + MSE, without leaving the device.  With mse_out =
 * NULL the call only enqueues (asynchronous); vsom_get_mse then returns the chunk's MSE.     */
int vsom_train_online_chunk(vsom_ctx *ctx, double eta, double sigma, int decay_fn,
                            float *mse_out);
/* The same with the epoch's MSE accumulator carried over: the reference keeps ONE running float over
 * all chunks of an epoch (declared before the chunk loop, Som.cpp:1153; every sample adds its
 * squaredNorm/epochSize, :1167).  first_chunk != 0 starts it at 0, otherwise it continues from the
 * previous chunk; mse_out / vsom_get_mse give the running value after this chunk.
 * vsom_train_online_chunk = first_chunk 1. */
int vsom_train_online_chunk_acc(vsom_ctx *ctx, double eta, double sigma, int decay_fn, int first_chunk,
                                float *mse_out);
/* The same as ONE synchronising call that also hands back what Som::trainBasicSom reads after the sample loop of an
 * epoch's last chunk: the chunk's lastBMU (trainSingle writes data.getLastBMU(s) as it goes, Som.cpp:895,1163) and the
 * running MSE (:1167,1175).  lastbmu_out: B values (NULL: not wanted); mse_out: the running value after this chunk
 * (NULL: not wanted).  Equivalent to vsom_train_online_chunk_acc(..., NULL) + vsom_get_last_bmu + vsom_get_mse; on
 * maps small enough for the one-launch chunk kernel the results come back through pinned memory the kernel itself
 * stores into -- one launch and one stream wait per chunk (the reference's own 10 x 10 x 9 scenario). */
int vsom_train_online_chunk_fetch(vsom_ctx *ctx, double eta, double sigma, int decay_fn, int first_chunk,
                                  uint64_t *lastbmu_out, float *mse_out);
/* diagnostics of the chunk loop's image-bounded search (csrc/vsom_online.hip; synchronises): out[0] = samples searched
 * through the image since the last reset, out[1] = nodes evaluated exactly for them (the candidates that survived the
 * bound), out[2] = refinement workgroups that had a candidate, out[3] = 0.  All zero while the exact scan is in use. */
int vsom_get_online_search_stats(vsom_ctx *ctx, uint64_t *out /*[4]*/, int reset);

/* ---- multi-GPU batch epoch, one process, the GPUs of one node ------------------------
 * Som::trainBatchSomEpoch's two loops shard differently: phase 1 (Som.cpp:764-782 / 786-805) is
 * independent per SAMPLE, phase 2 (Som.cpp:809-876) is independent per NODE but sequential in samples
 * (the variance accumulator uses the prefix mean).  A group holds one context per device, each with
 * the whole map and the whole chunk; an epoch runs phase 1 on the device's samples, all-gathers
 * lastBMU / ||residual||^2, forms bmuHits and the MSE on every device, runs phase 2 on the device's
 * nodes and all-gathers the new map rows (sigmaMap / weightMap rows follow on a second stream behind the
 * next search).  Every device ends with the bit-identical state of the single-GPU epoch.
 * Transport: RCCL over xGMI (librccl is bound at run time; ncclCommInitAll over the devices);
 * VSOM_GROUP_TRANSPORT=peer, or a device list that repeats a device (rehearsal of the N > 1 flow on a
 * one-GPU box, which RCCL refuses), uses ordered peer copies instead.
 * devices = NULL means devices 0 .. ndev-1; ndev = 0 means every visible device.
 * vsom_group_ctx(g, r) exposes a member for the single-context calls (searches, getters): call
 * vsom_group_synchronize first, and write state only through vsom_group_set_state.               */
typedef struct vsom_group vsom_group;
int vsom_group_create(vsom_group **out, int ndev, const int *devices, uint32_t width, uint32_t height,
                      uint32_t in_len, int transform);
void vsom_group_destroy(vsom_group *g);
int vsom_group_size(const vsom_group *g);
vsom_ctx *vsom_group_ctx(vsom_group *g, int rank);
const char *vsom_group_transport(const vsom_group *g);   /* "rccl" or "peer" */
int vsom_group_synchronize(vsom_group *g);
int vsom_group_set_state(vsom_group *g, const float *map, const float *sigma, const float *S,
                         const float *weight, const uint64_t *bmu_hits);
int vsom_group_get_state(vsom_group *g, float *map, float *sigma, float *S, float *weight,
                         uint64_t *bmu_hits);
int vsom_group_set_update_mode(vsom_group *g, int mode);
int vsom_group_set_bmu_mode(vsom_group *g, int mode);
/* DataSet::loadNextDataFromStream for the group: every device copies ITS 1/n of the rows from the host
 * and the rest arrives by all-gather; upload = prefetch + commit + wait (same contract as the
 * single-context calls above) */
int vsom_group_upload_chunk(vsom_group *g, const float *x_host, size_t B);
int vsom_group_prefetch_chunk(vsom_group *g, const float *x_host, size_t B);
int vsom_group_prefetch_wait(vsom_group *g);
int vsom_group_commit_chunk(vsom_group *g);
/* DataSet::loadNextDataFromStream (DataSet.cpp:118-160) for a chunk already resident in HBM: rows_dev[r] = member r's own rows [B*r/n, B*(r+1)/n) on ITS
 * device; all-gather of the rows, staging, lastBMU := 0 -- asynchronous on the members' streams */
int vsom_group_set_chunk_device(vsom_group *g, const void *const *rows_dev /*[n]*/, size_t B);
int vsom_group_set_last_bmu(vsom_group *g, const uint64_t *in_host);
int vsom_group_get_last_bmu(vsom_group *g, uint64_t *out_host);
/* Som::trainBatchSomEpoch (Som.cpp:756-879) over the group */
int vsom_group_batch_epoch_async(vsom_group *g, double sigma, int is_first);
int vsom_group_batch_epoch(vsom_group *g, double sigma, int is_first, float *mse_out);
int vsom_group_get_mse(vsom_group *g, float *mse_out);

/* ---- static helper: Som::calculateNeighbourhoodWeight (Som.cpp:949-975) ---------------- */
double vsom_neighbourhood_weight(size_t cx, size_t cy, size_t bx, size_t by, double sigma);

/* ---- interop / measurement -------------------------------------------------------------*/
/* raw device pointer of a context buffer (for RCCL / torch.distributed collectives) */
void *vsom_device_ptr(vsom_ctx *ctx, int which);
size_t vsom_chunk_size(const vsom_ctx *ctx);
/* row pitch in floats of the MAP/SIGMA/S device buffers (>= D; CLR: [A | pad | B | pad]) and
 * of the staged CHUNK buffer */
uint32_t vsom_pitch(const vsom_ctx *ctx);
uint32_t vsom_chunk_pitch(const vsom_ctx *ctx);
/* 1 when phase 2 over a shard of `nodes` nodes takes the small-map chain kernel (lane = (node, dim pair),
   update_chain3_kernel) instead of the lane = node kernels -- for reports that name the dominant kernel */
int vsom_small_map_chains(const vsom_ctx *ctx, size_t nodes);
/* per-kernel-group HIP-event timing on the context stream */
int vsom_enable_timing(vsom_ctx *ctx, int on);
/* the same for the groups whose bit (1u << VSOM_T_*) is set only: every timed group puts two event records between
 * kernels that otherwise run back to back (~5 us of idle device each on this chip), so a measurement of a whole step
 * times the one group it needs */
int vsom_enable_timing_of(vsom_ctx *ctx, uint32_t group_mask);
/* accumulated milliseconds and launch counts since the last reset (synchronises) */
int vsom_get_timing(vsom_ctx *ctx, float *ms_out /*[VSOM_T_COUNT]*/,
                    uint32_t *count_out /*[VSOM_T_COUNT]*/, int reset);

#ifdef __cplusplus
}
#endif
#endif
