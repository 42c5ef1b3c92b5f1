// vsom_capi.hip -- the extern "C" entry points declared in include/vsom_hip.h.
#include "vsom_internal.hpp"

#include <cstring>
#include <cstdlib>
#include <cmath>
#include <new>

static thread_local std::string g_last_error;

void vsom_set_error(const std::string &msg) { g_last_error = msg; }
int vsom_fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

TimerScope::TimerScope(vsom_ctx *ctx, int w) : c(ctx), which(w), on((ctx->timing >> w) & 1u)
{
    if (!on)
        return;
    if (!c->ev_pool.empty()) {
        ev = c->ev_pool.back();
        c->ev_pool.pop_back();
    } else {
        if (hipEventCreate(&ev.a) != hipSuccess || hipEventCreate(&ev.b) != hipSuccess) {
            on = false;
            return;
        }
    }
    ev.which = which;
    (void)hipEventRecord(ev.a, c->stream);
}
TimerScope::~TimerScope()
{
    if (!on)
        return;
    (void)hipEventRecord(ev.b, c->stream);
    c->ev_live.push_back(ev);
}

static inline uint32_t roundup(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

#define CHECK_CTX(ctx)                                                 \
    do {                                                               \
        if (!(ctx))                                                    \
            return vsom_fail(VSOM_ERR_INVALID, "null context");        \
        hipError_t _e = hipSetDevice((ctx)->device);                   \
        if (_e != hipSuccess)                                          \
            return vsom_fail(VSOM_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(_e)); \
        if (int _rc = vsom_join_aux(ctx))                              \
            return _rc;                                                \
        (ctx)->rows_free_valid = false;   /* whatever follows may read the staged rows again */ \
    } while (0)
// Entry points that READ the staged rows (Xs / XP / YP, the gathered rows, the int8 images): once the NEXT chunk has been
// staged ahead (vsom_prefetch_chunk / vsom_stage_next_device beside a running epoch) those buffers hold the next chunk's
// rows while B, lastBMU and the compaction record still describe the current one -- a search would silently mix the two.
// (ahead_rows, not ahead_valid: a staged-ahead chunk that was abandoned for another one has overwritten the rows all the same.)
#define CHECK_ROWS(ctx)                                                \
    do {                                                               \
        if ((ctx)->ahead_rows)                                         \
            return vsom_fail(VSOM_ERR_INVALID,                         \
                             "the next chunk is staged ahead over the current chunk's rows: vsom_commit_chunk first"); \
    } while (0)
// phase 2 runs beside the side stream's work and joins it at its end
#define CHECK_CTX_NOJOIN(ctx)                                          \
    do {                                                               \
        if (!(ctx))                                                    \
            return vsom_fail(VSOM_ERR_INVALID, "null context");        \
        hipError_t _e = hipSetDevice((ctx)->device);                   \
        if (_e != hipSuccess)                                          \
            return vsom_fail(VSOM_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(_e)); \
    } while (0)

// ---- pieces of the double-buffered ingest shared with the multi-GPU group (vsom_group.hip) -------------
// Rows [r0,r1) of a B-row host chunk -> the same rows of the context's NEXT raw device buffer, on the
// copy stream (a group copies each device's shard only and all-gathers the rest over xGMI).
int vsom_prefetch_rows(vsom_ctx *c, const float *x_host, size_t B, size_t r0, size_t r1)
{
    CHECK_CTX_NOJOIN(c);
    if (B > 0 && !x_host)
        return vsom_fail(VSOM_ERR_INVALID, "x_host is null");
    if (B > 0x7FFFFFFFull)
        return vsom_fail(VSOM_ERR_INVALID, "chunk too large");
    if (r0 > r1 || r1 > B)
        return vsom_fail(VSOM_ERR_INVALID, "row range out of bounds");
    const int k = c->next_slot;
    const size_t need = B * c->J;
    // the staging kernels of the chunk committed from this slot two prefetches ago must be done
    if (c->staged_valid[k])
        VSOM_HIP_CHECK(hipStreamWaitEvent(c->copy_stream, c->ev_staged[k], 0));
    if (need > c->Xnext_cap[k]) {
        if (c->staged_valid[k])
            VSOM_HIP_CHECK(hipEventSynchronize(c->ev_staged[k]));
        VSOM_HIP_CHECK(hipStreamSynchronize(c->copy_stream));
        if (c->Xnext[k])
            (void)hipFree(c->Xnext[k]);
        c->Xnext[k] = nullptr;
        c->Xnext_cap[k] = 0;
        VSOM_HIP_CHECK(hipMalloc(&c->Xnext[k], need * 4));
        c->Xnext_cap[k] = need;
    }
    if (r1 > r0)
        VSOM_HIP_CHECK(hipMemcpyAsync(c->Xnext[k] + r0 * c->J, x_host + r0 * c->J, (r1 - r0) * c->J * 4,
                                      hipMemcpyHostToDevice, c->copy_stream));
    VSOM_HIP_CHECK(hipEventRecord(c->ev_copied[k], c->copy_stream));
    c->Bnext = B;
    c->ready_slot = k;
    c->next_slot = k ^ 1;
    return VSOM_OK;
}

// first half of vsom_commit_chunk: the compute stream waits for the copy; *raw = the B x J rows
int vsom_commit_begin(vsom_ctx *c, float **raw, size_t *B)
{
    CHECK_CTX(c);
    if (c->ready_slot < 0)
        return vsom_fail(VSOM_ERR_INVALID, "no prefetched chunk to commit");
    const int k = c->ready_slot;
    VSOM_HIP_CHECK(hipStreamWaitEvent(c->stream, c->ev_copied[k], 0));
    *raw = c->Xnext[k];
    *B = c->Bnext;
    return VSOM_OK;
}

// second half: stage the rows (lastBMU := 0) and mark the slot reusable once staging has read it
int vsom_commit_end(vsom_ctx *c)
{
    CHECK_CTX(c);
    if (c->ready_slot < 0)
        return vsom_fail(VSOM_ERR_INVALID, "no prefetched chunk to commit");
    const int k = c->ready_slot;
    c->ready_slot = -1;
    if (c->ahead_valid)           // staged beside the previous epoch (vsom_prefetch_chunk): nothing left to launch
        return vsom_adopt_ahead(c);
    int rc = vsom_set_chunk_device(c, c->Xnext[k], c->Bnext);
    if (rc)
        return rc;
    VSOM_HIP_CHECK(hipEventRecord(c->ev_staged[k], c->stream));
    c->staged_valid[k] = true;
    return VSOM_OK;
}

extern "C" {

const char *vsom_last_error(void) { return g_last_error.c_str(); }

int vsom_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        return 0;
    return n;
}

static int free_all(vsom_ctx *c)
{
    void *ptrs[] = {c->map, c->sigma, c->S, c->weight, c->hits, c->Xs, c->XP, c->YP, c->Xraw,
                    c->lastbmu, c->sqres, c->pair_i, c->pair_j, c->partial, c->nan0,
                    c->cw, c->lut, c->lutd, c->sl_G, c->sl_nrm, c->sl_scal, c->sl_list, c->sl_tmin, c->sl_fs, c->sl_fm, c->v_dev, c->res_dev, c->onl_state, c->onl_f,
                    c->cc_flags, c->cc_idx, c->cc_inv, c->cc_meta, c->Xc, c->Mc, c->Uc_map, c->Uc_S, c->Xq, c->zq, c->sl_xi, c->sl_l1, c->sl_q, c->sl_qscale, c->sl_qcorr,
                    c->lastbmu_alt, c->cc_idx_alt, c->cc_inv_alt, c->cc_meta_alt, c->sl_a2, c->sl_qfast,
                    c->onl_img, c->onl_nsc, c->onl_lb, c->onl_u, c->onl_xsc, c->q_scratch, c->onl_dirty, c->dd_hash, c->dd_rep, c->dd_list};
    for (void *p : ptrs)
        if (p)
            (void)hipFree(p);
    if (c->lut_host)
        (void)hipHostFree(c->lut_host);
    for (int i = 0; i < 2; ++i)
        if (c->lut_ev[i])
            (void)hipEventDestroy(c->lut_ev[i]);
    if (c->v_pinned)
        (void)hipHostFree(c->v_pinned);
    if (c->st_pinned)
        (void)hipHostFree(c->st_pinned);
    if (c->mse)
        (void)hipHostFree(c->mse);
    if (c->lutd_host)
        (void)hipHostFree(c->lutd_host);
    for (int i = 0; i < 2; ++i)
        if (c->lutd_ev[i])
            (void)hipEventDestroy(c->lutd_ev[i]);
    if (c->out_pinned)
        (void)hipHostFree(c->out_pinned);
    if (c->sl_fb)
        (void)hipHostFree(c->sl_fb);
    if (c->cc_fb)
        (void)hipHostFree(c->cc_fb);
    for (auto &e : c->ev_live) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    for (auto &e : c->ev_pool) {
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    if (c->upd_module)
        (void)hipModuleUnload((hipModule_t)c->upd_module);
    if (c->aux_stream) {
        (void)hipStreamSynchronize(c->aux_stream);
        (void)hipStreamDestroy(c->aux_stream);
    }
    if (c->copy_stream) {
        (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamDestroy(c->copy_stream);
    }
    for (int i = 0; i < 2; ++i) {
        if (c->Xnext[i])
            (void)hipFree(c->Xnext[i]);
        if (c->ev_copied[i])
            (void)hipEventDestroy(c->ev_copied[i]);
        if (c->ev_staged[i])
            (void)hipEventDestroy(c->ev_staged[i]);
    }
    if (c->ev_rows_free)
        (void)hipEventDestroy(c->ev_rows_free);
    if (c->ev_ahead)
        (void)hipEventDestroy(c->ev_ahead);
    if (c->ev_fork)
        (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join)
        (void)hipEventDestroy(c->ev_join);
    if (c->own_stream)
        (void)hipStreamDestroy(c->own_stream);
    return 0;
}

int vsom_create(vsom_ctx **out, int device, uint32_t width, uint32_t height, uint32_t in_len,
                int transform)
{
    if (!out)
        return vsom_fail(VSOM_ERR_INVALID, "out is null");
    *out = nullptr;
    if (width == 0 || height == 0 || in_len == 0)
        return vsom_fail(VSOM_ERR_INVALID, "width, height and in_len must be > 0");
    if (transform < VSOM_STANDARD || transform > VSOM_CLR)
        return vsom_fail(VSOM_ERR_INVALID, "unknown transformation kind");
    if (transform == VSOM_CLR && in_len < 2)
        return vsom_fail(VSOM_ERR_INVALID, "CombinatorialLinearRegression needs in_len >= 2");
    if ((uint64_t)width * height > 0x7FFFFFFFull)
        return vsom_fail(VSOM_ERR_INVALID, "map too large");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return vsom_fail(VSOM_ERR_HIP, "no HIP device available (libvsom_hip has no CPU fallback)");
    if (device < 0 || device >= ndev)
        return vsom_fail(VSOM_ERR_INVALID, "device index out of range");
    VSOM_HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    VSOM_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return vsom_fail(VSOM_ERR_UNSUPPORTED,
                         std::string("libvsom_hip is built for gfx950 only, device is ") + prop.gcnArchName);

    vsom_ctx *c = new (std::nothrow) vsom_ctx();
    if (!c)
        return vsom_fail(VSOM_ERR_NOMEM, "out of host memory");
    c->device = device;
    c->W = width;
    c->H = height;
    c->J = in_len;
    c->N = width * height;
    c->transform = transform;
    // Transformation::Length (Transformation.cpp:31-35, 69-73, 162-165)
    c->D = transform == VSOM_CLR ? in_len * (in_len - 1u) : in_len;
    c->nparts = transform == VSOM_CLR ? 2 : 1;
    c->part_len = c->D / c->nparts;
    c->part_pitch = roundup(c->part_len, VSOM_TK);
    c->pitch = c->nparts * c->part_pitch;
    c->xpitch = roundup(c->J, VSOM_TK);
    if (const char *e = std::getenv("VSOM_NO_TINY"))
        c->use_tiny = !(e[0] == '1');
    if (const char *e = std::getenv("VSOM_COMPACT_MIN_ROWS"))     // development: initial vsom_set_column_compaction
        c->cc_min_rows = std::atol(e);
    if (const char *e = std::getenv("VSOM_NO_DEDUPE"))
        c->dedupe = !(e[0] == '1');     // A/B measurements of the duplicate-row pass of the exact search
    if (const char *e = std::getenv("VSOM_NO_CHAIN"))
        c->use_chain = !(e[0] == '1');  // debugging aid: lane = node update kernel on small maps too

    int rc = VSOM_OK;
    do {
        // the copy stream also runs the staging kernels of a chunk staged AHEAD, beside the chains of the current one
        // (vsom_prefetch_chunk): lowest priority, so that the dispatcher hands them the slots the chain kernel leaves
        // free at its ragged end instead of taking turns with it (at equal priority the chains of a 128x128 map lost
        // 0.09 ms to 0.05 ms of staging kernels: tools/exp/ab_stage.py)
        int prio_low = 0, prio_high = 0;
        (void)hipDeviceGetStreamPriorityRange(&prio_low, &prio_high);
        if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess ||
            hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
            hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, prio_low) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_copied[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_copied[1], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_staged[0], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_staged[1], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_rows_free, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->ev_ahead, hipEventDisableTiming) != hipSuccess) {
            rc = vsom_fail(VSOM_ERR_HIP, "hipStreamCreate failed");
            break;
        }
        c->stream = c->own_stream;
        const size_t nd = (size_t)c->N * c->pitch;
        if (hipMalloc(&c->map, nd * 4) != hipSuccess || hipMalloc(&c->sigma, nd * 4) != hipSuccess ||
            hipMalloc(&c->S, nd * 4) != hipSuccess || hipMalloc(&c->weight, (size_t)c->N * 4) != hipSuccess ||
            hipMalloc(&c->hits, (size_t)c->N * 8) != hipSuccess || hipHostMalloc(&c->mse, 16) != hipSuccess ||
            hipMalloc(&c->onl_state, VSOM_ONL_STATE_BYTES) != hipSuccess || hipMalloc(&c->onl_f, 64) != hipSuccess) {
            rc = vsom_fail(VSOM_ERR_NOMEM, "hipMalloc of model state failed");
            break;
        }
        (void)hipMemsetAsync(c->map, 0, nd * 4, c->stream);
        (void)hipMemsetAsync(c->sigma, 0, nd * 4, c->stream);
        (void)hipMemsetAsync(c->S, 0, nd * 4, c->stream);
        (void)hipMemsetAsync(c->weight, 0, (size_t)c->N * 4, c->stream);
        (void)hipMemsetAsync(c->hits, 0, (size_t)c->N * 8, c->stream);
        std::memset(c->mse, 0, 16);       // (pinned host memory the kernels write through: vsom_get_mse reads it after a stream wait)
        if (transform == VSOM_CLR) {
            // pair tables, i<j lexicographic (Transformation.cpp:94-101; tests/test1.cpp:18-43)
            std::vector<int> pi(c->part_len), pj(c->part_len);
            size_t p = 0;
            for (uint32_t i = 0; i < in_len; ++i)
                for (uint32_t j = i + 1; j < in_len; ++j) {
                    pi[p] = (int)i;
                    pj[p] = (int)j;
                    ++p;
                }
            if (hipMalloc(&c->pair_i, p * 4) != hipSuccess || hipMalloc(&c->pair_j, p * 4) != hipSuccess) {
                rc = vsom_fail(VSOM_ERR_NOMEM, "hipMalloc of pair tables failed");
                break;
            }
            (void)hipMemcpy(c->pair_i, pi.data(), p * 4, hipMemcpyHostToDevice);
            (void)hipMemcpy(c->pair_j, pj.data(), p * 4, hipMemcpyHostToDevice);
        }
        if (hipStreamSynchronize(c->stream) != hipSuccess) {
            rc = vsom_fail(VSOM_ERR_HIP, "initialisation failed");
            break;
        }
    } while (0);
    if (rc != VSOM_OK) {
        std::string keep = g_last_error;
        free_all(c);
        delete c;
        g_last_error = keep;
        return rc;
    }
    *out = c;
    return VSOM_OK;
}

void vsom_destroy(vsom_ctx *c)
{
    if (!c)
        return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->aux_stream)
        (void)hipStreamSynchronize(c->aux_stream);
    free_all(c);
    delete c;
}

int vsom_set_stream(vsom_ctx *c, void *hip_stream)
{
    CHECK_CTX(c);
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return VSOM_OK;
}

int vsom_synchronize(vsom_ctx *c)
{
    CHECK_CTX(c);
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VSOM_OK;
}

int vsom_set_bmu_mode(vsom_ctx *c, int mode)
{
    if (!c || mode < VSOM_BMU_AUTO || mode > VSOM_BMU_SHORTLIST)
        return vsom_fail(VSOM_ERR_INVALID, "bad bmu mode");
    c->bmu_mode = mode;
    return VSOM_OK;
}

int vsom_get_shortlist_stats(vsom_ctx *c, uint32_t *out)
{
    CHECK_CTX(c);
    if (!out)
        return vsom_fail(VSOM_ERR_INVALID, "null output");
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 4; ++i)
        out[i] = c->sl_fb ? ((volatile unsigned *)c->sl_fb)[i] : 0u;
    return VSOM_OK;
}

int vsom_set_column_compaction(vsom_ctx *c, long min_rows)
{
    if (!c)
        return vsom_fail(VSOM_ERR_INVALID, "null context");
    c->cc_min_rows = min_rows;
    c->cc_skip = 0;
    return VSOM_OK;
}

int vsom_set_row_dedupe(vsom_ctx *c, double min_work)
{
    if (!c)
        return vsom_fail(VSOM_ERR_INVALID, "null context");
    c->dd_min_work = min_work;
    return VSOM_OK;
}

int vsom_set_update_mode(vsom_ctx *c, int mode)
{
    if (!c || (mode != VSOM_UPDATE_STRICT && mode != VSOM_UPDATE_FMA && mode != VSOM_UPDATE_FMA_SIGMA))
        return vsom_fail(VSOM_ERR_INVALID, "bad update mode");
    c->update_mode = mode;
    return VSOM_OK;
}

uint32_t vsom_depth(const vsom_ctx *c) { return c ? c->D : 0; }
uint32_t vsom_nodes(const vsom_ctx *c) { return c ? c->N : 0; }
uint32_t vsom_residual_len(const vsom_ctx *c) { return c ? c->part_len : 0; }
size_t vsom_chunk_size(const vsom_ctx *c) { return c ? c->B : 0; }

static bool all_zero_bits(const void *p, size_t bytes)
{
    const uint64_t *w = static_cast<const uint64_t *>(p);
    size_t i = 0;
    for (; i + 8 <= bytes / 8; i += 8)
        if (w[i] | w[i + 1] | w[i + 2] | w[i + 3] | w[i + 4] | w[i + 5] | w[i + 6] | w[i + 7])
            return false;
    for (size_t b = i * 8; b < bytes; ++b)
        if (static_cast<const unsigned char *>(p)[b])
            return false;
    return true;
}

// host [N][D] <-> device [N][pitch] (each part separately for CLR).  Host -> device: an all-zero array (what
// Som::randomInitialize hands over for sigmaMap / SMap, Som.cpp:977-997) is a device fill; anything else travels through a
// pinned staging buffer -- a strided copy out of pageable memory took 5 ms per 4 MB array (tests/perf/ref_harness.py).
static int copy_rows(vsom_ctx *c, float *dev, const float *host_in, float *host_out)
{
    const size_t bytes = (size_t)c->N * c->D * 4;
    if (host_in && all_zero_bits(host_in, bytes)) {
        VSOM_HIP_CHECK(hipMemsetAsync(dev, 0, (size_t)c->N * c->pitch * 4, c->stream));   // (pad columns are zero anyway)
        return VSOM_OK;
    }
    if (host_in && bytes <= ((size_t)256 << 20)) {
        if (bytes > c->st_pinned_cap) {
            VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (c->st_pinned)
                (void)hipHostFree(c->st_pinned);
            c->st_pinned = nullptr;
            c->st_pinned_cap = 0;
            if (hipHostMalloc(&c->st_pinned, bytes) == hipSuccess)
                c->st_pinned_cap = bytes;
            else
                (void)hipGetLastError();
        }
        if (c->st_pinned_cap >= bytes) {
            VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));       // the previous array's copy out of the buffer
            std::memcpy(c->st_pinned, host_in, bytes);
            host_in = static_cast<const float *>(c->st_pinned);
        }
    }
    for (uint32_t part = 0; part < c->nparts; ++part) {
        float *d = dev + (size_t)part * c->part_pitch;
        if (host_in) {
            const float *h = host_in + (size_t)part * c->part_len;
            VSOM_HIP_CHECK(hipMemcpy2DAsync(d, (size_t)c->pitch * 4, h, (size_t)c->D * 4,
                                            (size_t)c->part_len * 4, c->N, hipMemcpyHostToDevice,
                                            c->stream));
        } else {
            float *h = host_out + (size_t)part * c->part_len;
            VSOM_HIP_CHECK(hipMemcpy2DAsync(h, (size_t)c->D * 4, d, (size_t)c->pitch * 4,
                                            (size_t)c->part_len * 4, c->N, hipMemcpyDeviceToHost,
                                            c->stream));
        }
    }
    return VSOM_OK;
}

int vsom_set_state(vsom_ctx *c, const float *map, const float *sigma, const float *S,
                   const float *weight, const uint64_t *bmu_hits)
{
    CHECK_CTX(c);
    int rc;
    if (map && (rc = copy_rows(c, c->map, map, nullptr)))
        return rc;
    if (sigma && (rc = copy_rows(c, c->sigma, sigma, nullptr)))
        return rc;
    if (S && (rc = copy_rows(c, c->S, S, nullptr)))
        return rc;
    if (weight) {
        if (all_zero_bits(weight, (size_t)c->N * 4))
            VSOM_HIP_CHECK(hipMemsetAsync(c->weight, 0, (size_t)c->N * 4, c->stream));
        else
            VSOM_HIP_CHECK(hipMemcpyAsync(c->weight, weight, (size_t)c->N * 4, hipMemcpyHostToDevice, c->stream));
    }
    if (bmu_hits) {
        if (all_zero_bits(bmu_hits, (size_t)c->N * 8))
            VSOM_HIP_CHECK(hipMemsetAsync(c->hits, 0, (size_t)c->N * 8, c->stream));
        else
            VSOM_HIP_CHECK(hipMemcpyAsync(c->hits, bmu_hits, (size_t)c->N * 8, hipMemcpyHostToDevice, c->stream));
    }
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VSOM_OK;
}

int vsom_get_state(vsom_ctx *c, float *map, float *sigma, float *S, float *weight, uint64_t *bmu_hits)
{
    CHECK_CTX(c);
    int rc;
    if (map && (rc = copy_rows(c, c->map, nullptr, map)))
        return rc;
    if (sigma && (rc = copy_rows(c, c->sigma, nullptr, sigma)))
        return rc;
    if (S && (rc = copy_rows(c, c->S, nullptr, S)))
        return rc;
    if (weight)
        VSOM_HIP_CHECK(hipMemcpyAsync(weight, c->weight, (size_t)c->N * 4, hipMemcpyDeviceToHost, c->stream));
    if (bmu_hits)
        VSOM_HIP_CHECK(hipMemcpyAsync(bmu_hits, c->hits, (size_t)c->N * 8, hipMemcpyDeviceToHost, c->stream));
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VSOM_OK;
}

static int ensure_chunk_capacity(vsom_ctx *c, size_t B)
{
    if (B <= c->Bcap)
        return VSOM_OK;
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    VSOM_HIP_CHECK(hipStreamSynchronize(c->copy_stream));
    c->ahead_valid = false;       // (a chunk staged ahead into the old buffers is staged anew at its commit)
    void **ptrs[] = {(void **)&c->Xs, (void **)&c->XP, (void **)&c->YP, (void **)&c->lastbmu, (void **)&c->lastbmu_alt,
                     (void **)&c->sqres, (void **)&c->nan0, (void **)&c->partial};
    for (void **p : ptrs) {
        if (*p)
            (void)hipFree(*p);
        *p = nullptr;
    }
    c->partial_cap = 0;
    c->Bcap = 0;
    // no chunk is staged from here on: should an allocation below fail, later entry points report
    // "no chunk loaded" instead of launching kernels on null buffers
    c->B = 0;
    c->chunk_loaded = false;
    size_t cap = (B + 63) / 64 * 64;
    // the fills below go to the context's stream: hipMemset would run on the null stream, which a
    // non-blocking stream does not wait for -- the fill could then land on top of the rows the staging
    // kernel writes next (seen as a few zero sample rows in one search of ~600 random cases)
    // the assembly update kernel reads up to 2 sample rows past the chunk and touches rows up to
    // PF_ROWS + 3 past it (gen_update_asm.py, load_cw)
    VSOM_HIP_CHECK(hipMalloc(&c->Xs, (cap + VSOM_ROW_PAD) * c->xpitch * 4));
    VSOM_HIP_CHECK(hipMemsetAsync(c->Xs, 0, (cap + VSOM_ROW_PAD) * c->xpitch * 4, c->stream));
    if (c->transform == VSOM_CLR) {
        // like Xs: the pipelined update kernels read one sample pair past the chunk
        VSOM_HIP_CHECK(hipMalloc(&c->XP, (cap + VSOM_ROW_PAD) * c->part_pitch * 4));
        VSOM_HIP_CHECK(hipMalloc(&c->YP, (cap + VSOM_ROW_PAD) * c->part_pitch * 4));
        VSOM_HIP_CHECK(hipMemsetAsync(c->XP, 0, (cap + VSOM_ROW_PAD) * c->part_pitch * 4, c->stream));
        VSOM_HIP_CHECK(hipMemsetAsync(c->YP, 0, (cap + VSOM_ROW_PAD) * c->part_pitch * 4, c->stream));
    }
    VSOM_HIP_CHECK(hipMalloc(&c->lastbmu, cap * 8));
    VSOM_HIP_CHECK(hipMalloc(&c->lastbmu_alt, cap * 8));
    VSOM_HIP_CHECK(hipMemsetAsync(c->lastbmu_alt, 0, cap * 8, c->stream));
    VSOM_HIP_CHECK(hipMalloc(&c->sqres, cap * 4));
    VSOM_HIP_CHECK(hipMalloc(&c->nan0, cap));
    VSOM_HIP_CHECK(hipMemsetAsync(c->lastbmu, 0, cap * 8, c->stream));
    VSOM_HIP_CHECK(hipMemsetAsync(c->sqres, 0, cap * 4, c->stream));
    VSOM_HIP_CHECK(hipMemsetAsync(c->nan0, 0, cap, c->stream));
    c->Bcap = cap;
    return VSOM_OK;
}

int vsom_set_chunk_device(vsom_ctx *c, const float *x_dev, size_t B)
{
    CHECK_CTX(c);
    if (B > 0 && !x_dev)
        return vsom_fail(VSOM_ERR_INVALID, "x_dev is null");
    if (B > 0x7FFFFFFFull)
        return vsom_fail(VSOM_ERR_INVALID, "chunk too large");
    int rc = ensure_chunk_capacity(c, B);
    if (rc)
        return rc;
    c->B = B;
    c->chunk_loaded = true;
    return launch_stage_chunk(c, x_dev, B);
}

static int upload_chunk_impl(vsom_ctx *c, const float *x_host, size_t B, bool wait)
{
    CHECK_CTX(c);
    if (B > 0 && !x_host)
        return vsom_fail(VSOM_ERR_INVALID, "x_host is null");
    size_t need = B * c->J;
    if (need > c->Xraw_cap) {
        VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
        if (c->Xraw)
            (void)hipFree(c->Xraw);
        c->Xraw = nullptr;
        c->Xraw_cap = 0;
        VSOM_HIP_CHECK(hipMalloc(&c->Xraw, need * 4));
        c->Xraw_cap = need;
    }
    if (need)
        VSOM_HIP_CHECK(hipMemcpyAsync(c->Xraw, x_host, need * 4, hipMemcpyHostToDevice, c->stream));
    int rc = vsom_set_chunk_device(c, c->Xraw, B);
    if (rc)
        return rc;
    if (wait)
        VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));   // x_host may be reused by the caller
    return VSOM_OK;
}

int vsom_upload_chunk(vsom_ctx *c, const float *x_host, size_t B) { return upload_chunk_impl(c, x_host, B, true); }

// copy and staging enqueued on the context's stream, no wait: x_host (pinned: vsom_host_alloc) stays the caller's to keep
// unchanged until a call that synchronises the context has returned
int vsom_upload_chunk_async(vsom_ctx *c, const float *x_host, size_t B) { return upload_chunk_impl(c, x_host, B, false); }

int vsom_host_alloc(void **out, size_t bytes)
{
    if (!out)
        return vsom_fail(VSOM_ERR_INVALID, "null output");
    *out = nullptr;
    if (bytes == 0)
        return VSOM_OK;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return vsom_fail(VSOM_ERR_NOMEM, "hipHostMalloc failed");
    }
    return VSOM_OK;
}

int vsom_host_free(void *p)
{
    if (p)
        VSOM_HIP_CHECK(hipHostFree(p));
    return VSOM_OK;
}

// what follows the copy of a prefetched chunk (or a chunk that already lives in HBM): its staging kernels on the copy
// stream, beside the epoch of the current chunk -- when that is possible now (vsom_can_stage_ahead), else at commit
static int stage_ahead_if_possible(vsom_ctx *c, const float *x_dev, size_t B)
{
    c->ahead_valid = false;       // (an abandoned ahead staging may still be running: ahead_rows stays set)
    if (!vsom_can_stage_ahead(c, B))
        return VSOM_OK;
    return launch_stage_chunk_ahead(c, x_dev, B);
}

int vsom_prefetch_chunk(vsom_ctx *c, const float *x_host, size_t B)
{
    int rc = vsom_prefetch_rows(c, x_host, B, 0, B);
    if (rc)
        return rc;
    c->next_dev_pending = false;
    const int k = c->ready_slot;
    if ((rc = stage_ahead_if_possible(c, c->Xnext[k], B)))
        return rc;
    if (c->ahead_valid) {         // the slot's raw rows have been read once the ahead staging is through
        VSOM_HIP_CHECK(hipEventRecord(c->ev_staged[k], c->copy_stream));
        c->staged_valid[k] = true;
    }
    return VSOM_OK;
}

int vsom_stage_next_device(vsom_ctx *c, const float *x_dev, size_t B)
{
    CHECK_CTX_NOJOIN(c);
    if (B > 0 && !x_dev)
        return vsom_fail(VSOM_ERR_INVALID, "x_dev is null");
    if (B > 0x7FFFFFFFull)
        return vsom_fail(VSOM_ERR_INVALID, "chunk too large");
    c->ready_slot = -1;           // replaces a prefetched chunk that was never committed
    c->next_dev = x_dev;
    c->next_dev_B = B;
    c->next_dev_pending = true;
    return stage_ahead_if_possible(c, x_dev, B);
}

int vsom_prefetch_wait(vsom_ctx *c)
{
    CHECK_CTX_NOJOIN(c);
    VSOM_HIP_CHECK(hipStreamSynchronize(c->copy_stream));
    return VSOM_OK;
}

int vsom_commit_chunk(vsom_ctx *c)
{
    if (c && c->next_dev_pending) {       // vsom_stage_next_device: adopt what was staged ahead, or stage it now
        CHECK_CTX(c);
        c->next_dev_pending = false;
        if (c->ahead_valid)
            return vsom_adopt_ahead(c);
        return vsom_set_chunk_device(c, c->next_dev, c->next_dev_B);
    }
    float *raw = nullptr;
    size_t B = 0;
    int rc = vsom_commit_begin(c, &raw, &B);
    if (rc)
        return rc;
    return vsom_commit_end(c);
}

__global__ void copy_u64_kernel(u64 *dst, const u64 *src, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        dst[i] = src[i];
}

int vsom_get_last_bmu(vsom_ctx *c, uint64_t *out_host)
{
    CHECK_CTX(c);
    if (c->B && !out_host)
        return vsom_fail(VSOM_ERR_INVALID, "null output");
    // short chunks (the reference's own scenarios train 20 rows an epoch): one small kernel stores the indices into pinned
    // host memory -- a device-to-host copy into the caller's pageable buffer was 18 us beyond the wait for the chunk
    if (c->B && c->B <= 8192) {
        if (!c->out_pinned)
            VSOM_HIP_CHECK(hipHostMalloc(&c->out_pinned, 8192 * sizeof(uint64_t)));
        hipLaunchKernelGGL(copy_u64_kernel, dim3((unsigned)((c->B + 255) / 256)), dim3(256), 0, c->stream,
                           static_cast<u64 *>(c->out_pinned), c->lastbmu, (int)c->B);
        VSOM_HIP_CHECK(hipGetLastError());
        VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
        std::memcpy(out_host, c->out_pinned, c->B * 8);
        return VSOM_OK;
    }
    if (c->B)
        VSOM_HIP_CHECK(hipMemcpyAsync(out_host, c->lastbmu, c->B * 8, hipMemcpyDeviceToHost, c->stream));
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VSOM_OK;
}

int vsom_set_last_bmu(vsom_ctx *c, const uint64_t *in_host)
{
    CHECK_CTX(c);
    if (c->B && !in_host)
        return vsom_fail(VSOM_ERR_INVALID, "null input");
    for (size_t i = 0; i < c->B; ++i)
        if (in_host[i] >= c->N)
            return vsom_fail(VSOM_ERR_INVALID, "lastBMU index out of range");
    if (c->B)
        VSOM_HIP_CHECK(hipMemcpyAsync(c->lastbmu, in_host, c->B * 8, hipMemcpyHostToDevice, c->stream));
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VSOM_OK;
}

int vsom_get_sqres(vsom_ctx *c, float *out_host)
{
    CHECK_CTX(c);
    if (c->B && !out_host)
        return vsom_fail(VSOM_ERR_INVALID, "null output");
    if (c->B)
        VSOM_HIP_CHECK(hipMemcpyAsync(out_host, c->sqres, c->B * 4, hipMemcpyDeviceToHost, c->stream));
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VSOM_OK;
}

static int copy_search_results(vsom_ctx *c, uint64_t *idx, float *dist)
{
    if (idx && c->B)
        VSOM_HIP_CHECK(hipMemcpyAsync(idx, c->lastbmu, c->B * 8, hipMemcpyDeviceToHost, c->stream));
    if (dist && c->B)
        VSOM_HIP_CHECK(hipMemcpyAsync(dist, c->sqres, c->B * 4, hipMemcpyDeviceToHost, c->stream));
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VSOM_OK;
}

int vsom_bmu_batch(vsom_ctx *c, uint64_t *idx_out_host, float *dist_out_host)
{
    CHECK_CTX(c);
    CHECK_ROWS(c);
    int rc = launch_bmu_full(c, 0, c->B);
    if (rc)
        return rc;
    return copy_search_results(c, idx_out_host, dist_out_host);
}

int vsom_bmu_local_batch(vsom_ctx *c, uint64_t *idx_out_host, float *dist_out_host)
{
    CHECK_CTX(c);
    CHECK_ROWS(c);
    int rc = launch_bmu_local(c, 0, c->B);
    if (rc)
        return rc;
    return copy_search_results(c, idx_out_host, dist_out_host);
}

// device scratch of the distance queries (pair lists in, distances out): grow-only, kept with the context -- a
// hipMalloc / hipFree pair per call cost more than the queries' kernels (tests/perf/ref_harness.py)
static int ensure_query_scratch(vsom_ctx *c, size_t bytes)
{
    if (bytes <= c->q_scratch_cap)
        return VSOM_OK;
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->q_scratch)
        (void)hipFree(c->q_scratch);
    c->q_scratch = nullptr;
    c->q_scratch_cap = 0;
    const size_t cap = (bytes + 4095) / 4096 * 4096;
    VSOM_HIP_CHECK(hipMalloc(&c->q_scratch, cap));
    c->q_scratch_cap = cap;
    return VSOM_OK;
}

int vsom_distances(vsom_ctx *c, const uint64_t *nodes_host, const uint64_t *rows_host, size_t count,
                   float *dist_out_host)
{
    CHECK_CTX(c);
    CHECK_ROWS(c);
    if (count == 0)
        return VSOM_OK;
    if (!nodes_host || !rows_host || !dist_out_host)
        return vsom_fail(VSOM_ERR_INVALID, "null argument");
    if (count > 0x0FFFFFFFull)
        return vsom_fail(VSOM_ERR_INVALID, "too many pairs");
    for (size_t i = 0; i < count; ++i)
        if (nodes_host[i] >= c->N || rows_host[i] >= c->B)
            return vsom_fail(VSOM_ERR_INVALID, "pair index out of range");
    const size_t c8 = (count * 8 + 255) / 256 * 256;
    int rc = ensure_query_scratch(c, 2 * c8 + count * 4);
    if (rc)
        return rc;
    u64 *dn = reinterpret_cast<u64 *>(c->q_scratch), *dr = reinterpret_cast<u64 *>((char *)c->q_scratch + c8);
    float *dd = reinterpret_cast<float *>((char *)c->q_scratch + 2 * c8);
    VSOM_HIP_CHECK(hipMemcpyAsync(dn, nodes_host, count * 8, hipMemcpyHostToDevice, c->stream));
    VSOM_HIP_CHECK(hipMemcpyAsync(dr, rows_host, count * 8, hipMemcpyHostToDevice, c->stream));
    if ((rc = launch_pair_dist(c, dn, dr, count, dd)))
        return rc;
    VSOM_HIP_CHECK(hipMemcpyAsync(dist_out_host, dd, count * 4, hipMemcpyDeviceToHost, c->stream));
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VSOM_OK;
}

int vsom_bmu_restricted_batch(vsom_ctx *c, uint64_t min_hits, uint64_t *idx_out_host, float *dist_out_host)
{
    CHECK_CTX(c);
    CHECK_ROWS(c);
    if (c->B == 0)
        return vsom_fail(VSOM_ERR_INVALID, "no chunk loaded");
    int rc = launch_bmu_restricted(c, min_hits);
    if (rc)
        return rc;
    return copy_search_results(c, idx_out_host, dist_out_host);
}

int vsom_distances_row(vsom_ctx *c, size_t row, float *dist_out_host)
{
    CHECK_CTX(c);
    CHECK_ROWS(c);
    if (row >= c->B || !dist_out_host)
        return vsom_fail(VSOM_ERR_INVALID, "row out of range or null output");
    int rc = ensure_query_scratch(c, (size_t)c->N * 4);
    if (rc)
        return rc;
    float *dd = reinterpret_cast<float *>(c->q_scratch);
    if ((rc = launch_row_dist(c, row, dd)))
        return rc;
    VSOM_HIP_CHECK(hipMemcpyAsync(dist_out_host, dd, (size_t)c->N * 4, hipMemcpyDeviceToHost, c->stream));
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VSOM_OK;
}

int vsom_distances_raw(vsom_ctx *c, const uint64_t *nodes_host, const uint64_t *vrows_host, size_t count,
                       int from_map, float *dist_out_host)
{
    CHECK_CTX(c);
    if (!from_map)
        CHECK_ROWS(c);
    if (count == 0)
        return VSOM_OK;
    if (!nodes_host || !vrows_host || !dist_out_host)
        return vsom_fail(VSOM_ERR_INVALID, "null argument");
    for (size_t i = 0; i < count; ++i)
        if (nodes_host[i] >= c->N || vrows_host[i] >= (from_map ? (uint64_t)c->N : (uint64_t)c->B))
            return vsom_fail(VSOM_ERR_INVALID, "pair index out of range");
    const size_t c8 = (count * 8 + 255) / 256 * 256;
    int rc = ensure_query_scratch(c, 2 * c8 + count * 4);
    if (rc)
        return rc;
    u64 *dn = reinterpret_cast<u64 *>(c->q_scratch), *dr = reinterpret_cast<u64 *>((char *)c->q_scratch + c8);
    float *dd = reinterpret_cast<float *>((char *)c->q_scratch + 2 * c8);
    VSOM_HIP_CHECK(hipMemcpyAsync(dn, nodes_host, count * 8, hipMemcpyHostToDevice, c->stream));
    VSOM_HIP_CHECK(hipMemcpyAsync(dr, vrows_host, count * 8, hipMemcpyHostToDevice, c->stream));
    if ((rc = launch_raw_dist(c, dn, dr, count, from_map, dd)))
        return rc;
    VSOM_HIP_CHECK(hipMemcpyAsync(dist_out_host, dd, count * 4, hipMemcpyDeviceToHost, c->stream));
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VSOM_OK;
}

int vsom_batch_phase1_async(vsom_ctx *c, size_t s0, size_t s1, int is_first)
{
    CHECK_CTX(c);
    CHECK_ROWS(c);
    if (s0 > s1 || s1 > c->B)
        return vsom_fail(VSOM_ERR_INVALID, "sample range out of bounds");
    return is_first ? launch_bmu_full(c, s0, s1) : launch_bmu_local(c, s0, s1);
}

int vsom_batch_finish_async(vsom_ctx *c)
{
    CHECK_CTX(c);
    if (!c->chunk_loaded)   // an EMPTY chunk is legal: the reference's epoch then zeroes the map
        return vsom_fail(VSOM_ERR_INVALID, "no chunk loaded");
    return launch_finish(c);
}

int vsom_batch_phase2_async(vsom_ctx *c, double sigma, size_t n0, size_t n1)
{
    CHECK_CTX_NOJOIN(c);
    if (!c->xq_valid)             // (a further node range of the same epoch works on the transposed chunk it already has)
        CHECK_ROWS(c);
    if (n0 > n1 || n1 > c->N)
        return vsom_fail(VSOM_ERR_INVALID, "node range out of bounds");
    if (!c->chunk_loaded)   // an EMPTY chunk is legal: the reference's epoch then zeroes the map
        return vsom_fail(VSOM_ERR_INVALID, "no chunk loaded");
    return launch_phase2(c, sigma, n0, n1);
}

int vsom_batch_epoch_async(vsom_ctx *c, double sigma, int is_first)
{
    CHECK_CTX(c);
    CHECK_ROWS(c);
    if (!c->chunk_loaded)   // an EMPTY chunk is legal: the reference's epoch then zeroes the map
        return vsom_fail(VSOM_ERR_INVALID, "no chunk loaded");
    if (vsom_tiny_applies(c))
        return launch_tiny_epoch(c, sigma, is_first);   // tiny map: the whole epoch in one launch
    int rc = vsom_batch_phase1_async(c, 0, c->B, is_first);
    if (rc)
        return rc;
    if ((rc = launch_finish(c)))
        return rc;
    return launch_phase2(c, sigma, 0, c->N);
}

int vsom_get_mse(vsom_ctx *c, float *mse_out)
{
    CHECK_CTX(c);
    if (!mse_out)
        return vsom_fail(VSOM_ERR_INVALID, "null output");
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    *mse_out = *static_cast<volatile float *>(c->mse);     // pinned host memory: the kernels' store lands here (22 -> 2 us a call)
    return VSOM_OK;
}

int vsom_batch_epoch(vsom_ctx *c, double sigma, int is_first, float *mse_out)
{
    int rc = vsom_batch_epoch_async(c, sigma, is_first);
    if (rc)
        return rc;
    if (mse_out)
        return vsom_get_mse(c, mse_out);
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    return VSOM_OK;
}

void *vsom_device_ptr(vsom_ctx *c, int which)
{
    if (!c)
        return nullptr;
    switch (which) {
    case VSOM_BUF_MAP: return c->map;
    case VSOM_BUF_SIGMA: return c->sigma;
    case VSOM_BUF_S: return c->S;
    case VSOM_BUF_WEIGHT: return c->weight;
    case VSOM_BUF_HITS: return c->hits;
    case VSOM_BUF_LASTBMU: return c->lastbmu;
    case VSOM_BUF_SQRES: return c->sqres;
    case VSOM_BUF_CHUNK: return c->Xs;
    default: return nullptr;
    }
}

uint32_t vsom_pitch(const vsom_ctx *c) { return c ? c->pitch : 0; }
uint32_t vsom_chunk_pitch(const vsom_ctx *c) { return c ? c->xpitch : 0; }

int vsom_enable_timing(vsom_ctx *c, int on)
{
    if (!c)
        return vsom_fail(VSOM_ERR_INVALID, "null context");
    c->timing = on ? ~0u : 0u;
    return VSOM_OK;
}

int vsom_enable_timing_of(vsom_ctx *c, uint32_t group_mask)
{
    if (!c)
        return vsom_fail(VSOM_ERR_INVALID, "null context");
    c->timing = group_mask;
    return VSOM_OK;
}

int vsom_get_timing(vsom_ctx *c, float *ms_out, uint32_t *count_out, int reset)
{
    CHECK_CTX(c);
    VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
    for (auto &e : c->ev_live) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            c->t_ms[e.which] += ms;
            c->t_cnt[e.which] += 1;
        }
        c->ev_pool.push_back(e);
    }
    c->ev_live.clear();
    for (int i = 0; i < VSOM_T_COUNT; ++i) {
        if (ms_out)
            ms_out[i] = c->t_ms[i];
        if (count_out)
            count_out[i] = c->t_cnt[i];
        if (reset) {
            c->t_ms[i] = 0.f;
            c->t_cnt[i] = 0;
        }
    }
    return VSOM_OK;
}

}   // extern "C"
