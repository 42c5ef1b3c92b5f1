// vsom_group.hip -- Som::trainBatchSomEpoch (Som.cpp:756-879) over the GPUs of one node, one host
// process: the vsom_group_* entry points of include/vsom_hip.h.
//
// Exact-parity partitioning (DESIGN.md "Multi-GPU"; the same orchestration dist.py runs with one
// process per GPU over torch.distributed):
//   chunk   : every device copies ITS rows of the host chunk (1/n of the PCIe traffic each) and the rows
//             are all-gathered over xGMI -- phase 2 reads every sample;
//   phase 1 : samples sharded (Som.cpp:764-782 / 786-805 are independent per sample), map replicated;
//             all-gather of lastBMU (8 B/sample) and ||residual||^2 (4 B/sample), then every device
//             forms bmuHits and the fp32 MSE in sample order itself;
//   phase 2 : nodes sharded (Som.cpp:809-876 is independent per node; the variance accumulator uses the
//             prefix mean, so sample-sharded partial sums cannot reproduce it), all-gather of the new
//             map rows; sigmaMap / weightMap rows are gathered on a second stream behind the next
//             search, which does not read them.
// Every device ends each epoch with the whole, bit-identical state of the single-GPU epoch.
//
// Transport: RCCL (librccl, bound at run time so that the single-GPU library has no hard dependency on
// it): ncclCommInitAll over the group's devices, ncclAllGather (even shards) or a group of
// ncclBroadcast (ragged shards) on the contexts' streams.  VSOM_GROUP_TRANSPORT=peer -- and a device
// list that names one device more than once, which RCCL refuses (rehearsal of the N > 1 flow on a
// one-GPU box) -- selects plain peer copies (hipMemcpyAsync device-to-device, the direct "1/n to each of
// the n-1 peers" pattern xGMI's point-to-point links favour) ordered with events.
#include "vsom_internal.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <new>
#include <set>

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi g_rccl;

// binds librccl once per process; an RCCL another component (torch) already loaded is reused
int load_rccl()
{
    if (g_rccl.handle)
        return VSOM_OK;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)
        if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL)))
            break;
    if (!h)
        return vsom_fail(VSOM_ERR_UNSUPPORTED, std::string("librccl not loadable: ") + dlerror());
    RcclApi a;
    a.handle = h;
    a.CommInitAll = (decltype(a.CommInitAll))dlsym(h, "ncclCommInitAll");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(h, "ncclCommDestroy");
    a.AllGather = (decltype(a.AllGather))dlsym(h, "ncclAllGather");
    a.Broadcast = (decltype(a.Broadcast))dlsym(h, "ncclBroadcast");
    a.GroupStart = (decltype(a.GroupStart))dlsym(h, "ncclGroupStart");
    a.GroupEnd = (decltype(a.GroupEnd))dlsym(h, "ncclGroupEnd");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(h, "ncclGetErrorString");
    if (!a.CommInitAll || !a.CommDestroy || !a.AllGather || !a.Broadcast || !a.GroupStart || !a.GroupEnd ||
        !a.GetErrorString)
        return vsom_fail(VSOM_ERR_UNSUPPORTED, "librccl lacks an expected entry point");
    g_rccl = a;
    return VSOM_OK;
}

}   // namespace

struct vsom_group {
    int n = 0;
    std::vector<vsom_ctx *> ctx;
    std::vector<int> dev;
    bool use_rccl = false;
    std::vector<ncclComm_t> comm;
    // second stream per device for the gathers nothing waits for until the next phase 2 / state read
    std::vector<hipStream_t> gstream;
    std::vector<hipEvent_t> ev_p2;        // phase 2 + map gather enqueued on the main stream
    std::vector<hipEvent_t> ev_gathered;  // deferred sigmaMap / weightMap gathers finished
    std::vector<hipEvent_t> ev_ready, ev_done;   // peer transport: source data ready / copies drained
    bool deferred_pending = false;
};

#define VSOM_NCCL_CHECK(expr)                                                                          \
    do {                                                                                               \
        ncclResult_t _r = (expr);                                                                      \
        if (_r != ncclSuccess)                                                                         \
            return vsom_fail(VSOM_ERR_HIP, std::string(#expr) + ": " + g_rccl.GetErrorString(_r));     \
    } while (0)

static inline void shard(size_t total, int n, int r, size_t &lo, size_t &hi)
{
    lo = total * (size_t)r / (size_t)n;
    hi = total * (size_t)(r + 1) / (size_t)n;
}

// All devices end up with every device's contiguous row shard of the buffer at base[r] (rows of
// row_bytes bytes, `total` rows, shard r = rows [total*r/n, total*(r+1)/n)), in place, enqueued on st[r].
static int gather_rows(vsom_group *g, std::vector<char *> &base, size_t row_bytes, size_t total,
                       std::vector<hipStream_t> &st)
{
    const int n = g->n;
    if (total == 0 || row_bytes == 0)
        return VSOM_OK;
    if (g->use_rccl) {
        VSOM_NCCL_CHECK(g_rccl.GroupStart());
        if (total % (size_t)n == 0) {
            const size_t cnt = total / (size_t)n * row_bytes;
            for (int r = 0; r < n; ++r) {
                ncclResult_t rr = g_rccl.AllGather(base[r] + (size_t)r * cnt, base[r], cnt, ncclChar, g->comm[r], st[r]);
                if (rr != ncclSuccess) {
                    (void)g_rccl.GroupEnd();
                    return vsom_fail(VSOM_ERR_HIP, std::string("ncclAllGather: ") + g_rccl.GetErrorString(rr));
                }
            }
        } else {
            for (int src = 0; src < n; ++src) {
                size_t lo, hi;
                shard(total, n, src, lo, hi);
                if (hi <= lo)
                    continue;
                for (int r = 0; r < n; ++r) {
                    char *p = base[r] + lo * row_bytes;
                    ncclResult_t rr = g_rccl.Broadcast(p, p, (hi - lo) * row_bytes, ncclChar, src, g->comm[r], st[r]);
                    if (rr != ncclSuccess) {
                        (void)g_rccl.GroupEnd();
                        return vsom_fail(VSOM_ERR_HIP, std::string("ncclBroadcast: ") + g_rccl.GetErrorString(rr));
                    }
                }
            }
        }
        VSOM_NCCL_CHECK(g_rccl.GroupEnd());
        return VSOM_OK;
    }
    if (n == 1)
        return VSOM_OK;
    // peer transport: every destination stream waits until every source has produced its shard, pulls the
    // n-1 foreign shards, and no source goes on (it may overwrite its shard) before all pulls are done
    for (int r = 0; r < n; ++r) {
        VSOM_HIP_CHECK(hipSetDevice(g->dev[r]));
        VSOM_HIP_CHECK(hipEventRecord(g->ev_ready[r], st[r]));
    }
    for (int dst = 0; dst < n; ++dst) {
        VSOM_HIP_CHECK(hipSetDevice(g->dev[dst]));
        for (int src = 0; src < n; ++src) {
            if (src == dst)
                continue;
            size_t lo, hi;
            shard(total, n, src, lo, hi);
            if (hi <= lo)
                continue;
            VSOM_HIP_CHECK(hipStreamWaitEvent(st[dst], g->ev_ready[src], 0));
            VSOM_HIP_CHECK(hipMemcpyAsync(base[dst] + lo * row_bytes, base[src] + lo * row_bytes, (hi - lo) * row_bytes,
                                          hipMemcpyDeviceToDevice, st[dst]));
        }
        VSOM_HIP_CHECK(hipEventRecord(g->ev_done[dst], st[dst]));
    }
    for (int r = 0; r < n; ++r) {
        VSOM_HIP_CHECK(hipSetDevice(g->dev[r]));
        for (int o = 0; o < n; ++o)
            if (o != r)
                VSOM_HIP_CHECK(hipStreamWaitEvent(st[r], g->ev_done[o], 0));
    }
    return VSOM_OK;
}

static std::vector<hipStream_t> main_streams(vsom_group *g)
{
    std::vector<hipStream_t> s(g->n);
    for (int r = 0; r < g->n; ++r)
        s[r] = g->ctx[r]->stream;
    return s;
}

// the main streams wait for the deferred sigmaMap / weightMap gathers of the last epoch
static int join_deferred(vsom_group *g)
{
    if (!g->deferred_pending)
        return VSOM_OK;
    for (int r = 0; r < g->n; ++r) {
        VSOM_HIP_CHECK(hipSetDevice(g->dev[r]));
        VSOM_HIP_CHECK(hipStreamWaitEvent(g->ctx[r]->stream, g->ev_gathered[r], 0));
    }
    g->deferred_pending = false;
    return VSOM_OK;
}

extern "C" {

void vsom_group_destroy(vsom_group *g)
{
    if (!g)
        return;
    for (int r = 0; r < (int)g->ctx.size(); ++r) {
        (void)hipSetDevice(g->dev[r]);
        if (g->ctx[r])
            (void)hipStreamSynchronize(g->ctx[r]->stream);
        if (r < (int)g->gstream.size() && g->gstream[r])
            (void)hipStreamSynchronize(g->gstream[r]);
    }
    if (g->use_rccl)
        for (ncclComm_t c : g->comm)
            if (c)
                (void)g_rccl.CommDestroy(c);
    for (int r = 0; r < (int)g->ctx.size(); ++r) {
        (void)hipSetDevice(g->dev[r]);
        if (r < (int)g->gstream.size() && g->gstream[r])
            (void)hipStreamDestroy(g->gstream[r]);
        for (std::vector<hipEvent_t> *ev : {&g->ev_p2, &g->ev_gathered, &g->ev_ready, &g->ev_done})
            if (r < (int)ev->size() && (*ev)[r])
                (void)hipEventDestroy((*ev)[r]);
        vsom_destroy(g->ctx[r]);
    }
    delete g;
}

int vsom_group_create(vsom_group **out, int ndev, const int *devices, uint32_t width, uint32_t height,
                      uint32_t in_len, int transform)
{
    if (!out)
        return vsom_fail(VSOM_ERR_INVALID, "out is null");
    *out = nullptr;
    const int visible = vsom_device_count();
    if (ndev == 0)
        ndev = visible;   // all visible devices
    if (ndev <= 0)
        return vsom_fail(VSOM_ERR_HIP, "no HIP device available (libvsom_hip has no CPU fallback)");
    if (ndev > 64)
        return vsom_fail(VSOM_ERR_INVALID, "too many devices");
    vsom_group *g = new (std::nothrow) vsom_group();
    if (!g)
        return vsom_fail(VSOM_ERR_NOMEM, "out of host memory");
    g->n = ndev;
    std::set<int> distinct;
    for (int r = 0; r < ndev; ++r) {
        const int d = devices ? devices[r] : r;
        g->dev.push_back(d);
        distinct.insert(d);
    }
    int rc = VSOM_OK;
    for (int r = 0; r < ndev && rc == VSOM_OK; ++r) {
        vsom_ctx *c = nullptr;
        rc = vsom_create(&c, g->dev[r], width, height, in_len, transform);
        g->ctx.push_back(c);
    }
    auto fail = [&](int code) {
        std::string keep = vsom_last_error();
        vsom_group_destroy(g);
        vsom_set_error(keep);
        return code;
    };
    if (rc != VSOM_OK)
        return fail(rc);
    g->gstream.assign(ndev, nullptr);
    g->ev_p2.assign(ndev, nullptr);
    g->ev_gathered.assign(ndev, nullptr);
    g->ev_ready.assign(ndev, nullptr);
    g->ev_done.assign(ndev, nullptr);
    for (int r = 0; r < ndev; ++r) {
        if (hipSetDevice(g->dev[r]) != hipSuccess ||
            hipStreamCreateWithFlags(&g->gstream[r], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_p2[r], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_gathered[r], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_ready[r], hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&g->ev_done[r], hipEventDisableTiming) != hipSuccess) {
            vsom_fail(VSOM_ERR_HIP, "hipStreamCreate / hipEventCreate failed");
            return fail(VSOM_ERR_HIP);
        }
    }
    const char *tr = std::getenv("VSOM_GROUP_TRANSPORT");
    const bool want_peer = (tr && std::strcmp(tr, "peer") == 0) || (int)distinct.size() != ndev;
    if (!want_peer) {
        if ((rc = load_rccl()))
            return fail(rc);
        g->comm.assign(ndev, nullptr);
        ncclResult_t r = g_rccl.CommInitAll(g->comm.data(), ndev, g->dev.data());
        if (r != ncclSuccess) {
            vsom_fail(VSOM_ERR_HIP, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r));
            g->comm.clear();
            return fail(VSOM_ERR_HIP);
        }
        g->use_rccl = true;
    } else {
        // peer copies between distinct devices need peer access (xGMI); same-device "peers" do not
        for (int a = 0; a < ndev; ++a)
            for (int b = 0; b < ndev; ++b) {
                if (g->dev[a] == g->dev[b])
                    continue;
                (void)hipSetDevice(g->dev[a]);
                hipError_t e = hipDeviceEnablePeerAccess(g->dev[b], 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
                    vsom_fail(VSOM_ERR_HIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
                    return fail(VSOM_ERR_HIP);
                }
                (void)hipGetLastError();
            }
    }
    *out = g;
    return VSOM_OK;
}

int vsom_group_size(const vsom_group *g) { return g ? g->n : 0; }

vsom_ctx *vsom_group_ctx(vsom_group *g, int rank)
{
    if (!g || rank < 0 || rank >= g->n)
        return nullptr;
    return g->ctx[rank];
}

const char *vsom_group_transport(const vsom_group *g) { return !g ? "" : (g->use_rccl ? "rccl" : "peer"); }

int vsom_group_synchronize(vsom_group *g)
{
    if (!g)
        return vsom_fail(VSOM_ERR_INVALID, "null group");
    int rc = join_deferred(g);
    if (rc)
        return rc;
    for (int r = 0; r < g->n; ++r)
        if ((rc = vsom_synchronize(g->ctx[r])))
            return rc;
    return VSOM_OK;
}

int vsom_group_set_state(vsom_group *g, const float *map, const float *sigma, const float *S, const float *weight,
                         const uint64_t *bmu_hits)
{
    if (!g)
        return vsom_fail(VSOM_ERR_INVALID, "null group");
    int rc = join_deferred(g);
    for (int r = 0; r < g->n && rc == VSOM_OK; ++r)
        rc = vsom_set_state(g->ctx[r], map, sigma, S, weight, bmu_hits);
    return rc;
}

int vsom_group_get_state(vsom_group *g, float *map, float *sigma, float *S, float *weight, uint64_t *bmu_hits)
{
    if (!g)
        return vsom_fail(VSOM_ERR_INVALID, "null group");
    int rc = join_deferred(g);
    if (rc)
        return rc;
    return vsom_get_state(g->ctx[0], map, sigma, S, weight, bmu_hits);   // every device holds the whole state
}

int vsom_group_set_update_mode(vsom_group *g, int mode)
{
    if (!g)
        return vsom_fail(VSOM_ERR_INVALID, "null group");
    int rc = VSOM_OK;
    for (int r = 0; r < g->n && rc == VSOM_OK; ++r)
        rc = vsom_set_update_mode(g->ctx[r], mode);
    return rc;
}

int vsom_group_set_bmu_mode(vsom_group *g, int mode)
{
    if (!g)
        return vsom_fail(VSOM_ERR_INVALID, "null group");
    int rc = VSOM_OK;
    for (int r = 0; r < g->n && rc == VSOM_OK; ++r)
        rc = vsom_set_bmu_mode(g->ctx[r], mode);
    return rc;
}

int vsom_group_prefetch_chunk(vsom_group *g, const float *x_host, size_t B)
{
    if (!g)
        return vsom_fail(VSOM_ERR_INVALID, "null group");
    for (int r = 0; r < g->n; ++r) {
        size_t lo, hi;
        shard(B, g->n, r, lo, hi);
        int rc = vsom_prefetch_rows(g->ctx[r], x_host, B, lo, hi);   // this device's rows only
        if (rc)
            return rc;
    }
    return VSOM_OK;
}

int vsom_group_prefetch_wait(vsom_group *g)
{
    if (!g)
        return vsom_fail(VSOM_ERR_INVALID, "null group");
    for (int r = 0; r < g->n; ++r) {
        int rc = vsom_prefetch_wait(g->ctx[r]);
        if (rc)
            return rc;
    }
    return VSOM_OK;
}

int vsom_group_commit_chunk(vsom_group *g)
{
    if (!g)
        return vsom_fail(VSOM_ERR_INVALID, "null group");
    std::vector<char *> base(g->n);
    size_t B = 0;
    for (int r = 0; r < g->n; ++r) {
        float *raw = nullptr;
        size_t Br = 0;
        int rc = vsom_commit_begin(g->ctx[r], &raw, &Br);
        if (rc)
            return rc;
        if (r && Br != B)
            return vsom_fail(VSOM_ERR_INVALID, "group members hold prefetched chunks of different sizes");
        B = Br;
        base[r] = (char *)raw;
    }
    auto st = main_streams(g);
    int rc = gather_rows(g, base, (size_t)g->ctx[0]->J * 4, B, st);   // chunk replication over xGMI
    if (rc)
        return rc;
    for (int r = 0; r < g->n; ++r)
        if ((rc = vsom_commit_end(g->ctx[r])))
            return rc;
    return VSOM_OK;
}

// DataSet::loadNextDataFromStream for a chunk that already lives in HBM: rows_dev[r] = member r's OWN rows
// [B*r/n, B*(r+1)/n) on its device (contiguous, J floats each); the other rows arrive by all-gather over
// xGMI, then every member stages the whole chunk (lastBMU := 0).  Asynchronous on the members' streams; the
// caller's buffers may be reused after vsom_group_synchronize (or the next call that synchronises).
int vsom_group_set_chunk_device(vsom_group *g, const void *const *rows_dev, size_t B)
{
    if (!g || (B > 0 && !rows_dev))
        return vsom_fail(VSOM_ERR_INVALID, "null group / rows_dev");
    const int n = g->n;
    std::vector<char *> base(n);
    for (int r = 0; r < n; ++r) {
        vsom_ctx *c = g->ctx[r];
        VSOM_HIP_CHECK(hipSetDevice(g->dev[r]));
        const size_t need = B * c->J;
        if (need > c->Xraw_cap) {
            VSOM_HIP_CHECK(hipStreamSynchronize(c->stream));
            if (c->Xraw)
                (void)hipFree(c->Xraw);
            c->Xraw = nullptr;
            c->Xraw_cap = 0;
            VSOM_HIP_CHECK(hipMalloc(&c->Xraw, need * 4));
            c->Xraw_cap = need;
        }
        size_t lo, hi;
        shard(B, n, r, lo, hi);
        if (hi > lo) {
            if (!rows_dev[r])
                return vsom_fail(VSOM_ERR_INVALID, "rows_dev[r] is null");
            VSOM_HIP_CHECK(hipMemcpyAsync(c->Xraw + lo * c->J, rows_dev[r], (hi - lo) * (size_t)c->J * 4,
                                          hipMemcpyDeviceToDevice, c->stream));
        }
        base[r] = (char *)c->Xraw;
    }
    auto st = main_streams(g);
    int rc = gather_rows(g, base, (size_t)g->ctx[0]->J * 4, B, st);   // chunk replication over xGMI
    if (rc)
        return rc;
    for (int r = 0; r < n; ++r) {
        VSOM_HIP_CHECK(hipSetDevice(g->dev[r]));
        if ((rc = vsom_set_chunk_device(g->ctx[r], g->ctx[r]->Xraw, B)))
            return rc;
    }
    return VSOM_OK;
}

int vsom_group_upload_chunk(vsom_group *g, const float *x_host, size_t B)
{
    int rc = vsom_group_prefetch_chunk(g, x_host, B);
    if (rc)
        return rc;
    if ((rc = vsom_group_commit_chunk(g)))
        return rc;
    for (int r = 0; r < g->n; ++r)     // x_host may be reused by the caller
        if ((rc = vsom_synchronize(g->ctx[r])))
            return rc;
    return VSOM_OK;
}

int vsom_group_set_last_bmu(vsom_group *g, const uint64_t *in_host)
{
    if (!g)
        return vsom_fail(VSOM_ERR_INVALID, "null group");
    int rc = VSOM_OK;
    for (int r = 0; r < g->n && rc == VSOM_OK; ++r)
        rc = vsom_set_last_bmu(g->ctx[r], in_host);
    return rc;
}

int vsom_group_get_last_bmu(vsom_group *g, uint64_t *out_host)
{
    if (!g)
        return vsom_fail(VSOM_ERR_INVALID, "null group");
    return vsom_get_last_bmu(g->ctx[0], out_host);
}

int vsom_group_batch_epoch_async(vsom_group *g, double sigma, int is_first)
{
    if (!g)
        return vsom_fail(VSOM_ERR_INVALID, "null group");
    const int n = g->n;
    vsom_ctx *c0 = g->ctx[0];
    if (!c0->chunk_loaded)
        return vsom_fail(VSOM_ERR_INVALID, "no chunk loaded");
    const size_t B = c0->B, N = c0->N;
    for (int r = 1; r < n; ++r)
        if (!g->ctx[r]->chunk_loaded || g->ctx[r]->B != B)
            return vsom_fail(VSOM_ERR_INVALID, "group members hold different chunks");
    int rc;
    auto st = main_streams(g);
    std::vector<char *> base(n);
    // phase 1 on this device's samples
    for (int r = 0; r < n; ++r) {
        size_t lo, hi;
        shard(B, n, r, lo, hi);
        if ((rc = vsom_batch_phase1_async(g->ctx[r], lo, hi, is_first)))
            return rc;
    }
    for (int r = 0; r < n; ++r)
        base[r] = (char *)g->ctx[r]->lastbmu;
    if ((rc = gather_rows(g, base, 8, B, st)))
        return rc;
    for (int r = 0; r < n; ++r)
        base[r] = (char *)g->ctx[r]->sqres;
    if ((rc = gather_rows(g, base, 4, B, st)))
        return rc;
    for (int r = 0; r < n; ++r)
        if ((rc = vsom_batch_finish_async(g->ctx[r])))   // bmuHits, MSE in sample order: identical on every device
            return rc;
    // phase 2 rewrites the rows the last epoch's deferred gathers read
    if ((rc = join_deferred(g)))
        return rc;
    for (int r = 0; r < n; ++r) {
        size_t lo, hi;
        shard(N, n, r, lo, hi);
        if ((rc = vsom_batch_phase2_async(g->ctx[r], sigma, lo, hi)))
            return rc;
    }
    const size_t row = (size_t)c0->pitch * 4;
    for (int r = 0; r < n; ++r)
        base[r] = (char *)g->ctx[r]->map;
    if ((rc = gather_rows(g, base, row, N, st)))         // the next search needs the whole map
        return rc;
    if (n > 1 || g->use_rccl) {
        // sigmaMap / weightMap: behind the next search, on the second stream
        for (int r = 0; r < n; ++r) {
            VSOM_HIP_CHECK(hipSetDevice(g->dev[r]));
            VSOM_HIP_CHECK(hipEventRecord(g->ev_p2[r], st[r]));
            VSOM_HIP_CHECK(hipStreamWaitEvent(g->gstream[r], g->ev_p2[r], 0));
        }
        for (int r = 0; r < n; ++r)
            base[r] = (char *)g->ctx[r]->sigma;
        if ((rc = gather_rows(g, base, row, N, g->gstream)))
            return rc;
        for (int r = 0; r < n; ++r)
            base[r] = (char *)g->ctx[r]->weight;
        if ((rc = gather_rows(g, base, 4, N, g->gstream)))
            return rc;
        for (int r = 0; r < n; ++r) {
            VSOM_HIP_CHECK(hipSetDevice(g->dev[r]));
            VSOM_HIP_CHECK(hipEventRecord(g->ev_gathered[r], g->gstream[r]));
        }
        g->deferred_pending = true;
    }
    return VSOM_OK;
}

int vsom_group_get_mse(vsom_group *g, float *mse_out)
{
    if (!g)
        return vsom_fail(VSOM_ERR_INVALID, "null group");
    return vsom_get_mse(g->ctx[0], mse_out);
}

int vsom_group_batch_epoch(vsom_group *g, double sigma, int is_first, float *mse_out)
{
    int rc = vsom_group_batch_epoch_async(g, sigma, is_first);
    if (rc)
        return rc;
    if ((rc = vsom_group_synchronize(g)))
        return rc;
    if (mse_out)
        return vsom_get_mse(g->ctx[0], mse_out);
    return VSOM_OK;
}

}   // extern "C"
