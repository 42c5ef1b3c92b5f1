// vsom_host.cpp -- host-side implementation of vsom_api.hpp (libsom_hip.so).
//
// Every hot-path member of `Som` forwards to the C ABI of libvsom_hip.so (include/vsom_hip.h);
// the drivers (train / trainBatchSom / trainBasicSom) restate the reference's control flow
// (src/Som.cpp:716-754, 1113-1187) around those calls.  No training arithmetic runs here.
#include "vsom_api.hpp"
#include "vsom_checkpoint.hpp"
#include "../../../include/vsom_hip.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <fstream>
#include <iostream>
#include <stdexcept>
#include <random>
#include <ctime>

// ---------------------------------------------------------------------------------------------
// Transformation (reference src/Transformation.cpp:3-167).  Host evaluation of the built-in hooks
// is provided for callers that invoke transform.Comparer/Stepper directly; training uses the
// device kernels selected by kind().
// ---------------------------------------------------------------------------------------------
namespace vsom {

T StandardComparer::operator()(const T &value, const T &model, const T &, const T &) const
{
    T r(model.size());
    for (Eigen::Index i = 0; i < model.size(); ++i)
        r[i] = model[i] - value[i];   // model - value  (Transformation.cpp:8)
    return r;
}

T StandardStepper::operator()(const T &value, const T &model, const T &) const
{
    T r(model.size());
    for (Eigen::Index i = 0; i < model.size(); ++i)
        r[i] = value[i] - model[i];   // value - model  (Transformation.cpp:12)
    return r;
}

T MedianStepper::operator()(const T &value, const T &model, const T &) const
{
    T r(model.size());
    for (Eigen::Index i = 0; i < model.size(); ++i) {
        float a = value[i] - model[i];   // sign(value - model)  (Transformation.cpp:50)
        r[i] = (a != a) ? a : (float)((a > 0.f) - (a < 0.f));
    }
    return r;
}

static void clr_inner(const T &value, const T &model, T &inner, T &xprime)
{
    const Eigen::Index P = model.size() / 2;   // A = head, B = tail (Transformation.cpp:87-88)
    inner = T(P);
    xprime = T(P);
    Eigen::Index p = 0;
    for (Eigen::Index i = 0; i < value.size(); ++i)
        for (Eigen::Index j = i + 1; j < value.size(); ++j) {   // pairs i<j (Transformation.cpp:95-101)
            if (p < P) {
                float t = model[p] * value[i];
                t = t + model[P + p];
                t = t - value[j];
                inner[p] = t;
                xprime[p] = value[i];
            }
            ++p;
        }
}

T ClrComparer::operator()(const T &value, const T &model, const T &, const T &) const
{
    T inner, xp;
    clr_inner(value, model, inner, xp);
    return inner;   // A.*x' + B - y'  (Transformation.cpp:104)
}

T ClrStepper::operator()(const T &value, const T &model, const T &) const
{
    T inner, xp;
    clr_inner(value, model, inner, xp);
    const Eigen::Index P = inner.size();
    T delta(model.size());
    for (Eigen::Index p = 0; p < P; ++p) {
        float m2 = -2.f * inner[p];
        delta[p] = m2 * xp[p];    // aDelta (Transformation.cpp:135)
        delta[P + p] = m2;        // bDelta (Transformation.cpp:136)
    }
    return delta;
}

}   // namespace vsom

static std::function<std::vector<std::string>(const Eigen::VectorXf &)> make_displayer(std::vector<std::string> names, bool clr)
{
    return [names, clr](const Eigen::VectorXf &model) {
        std::vector<std::string> disp;
        if (!clr) {
            for (size_t i = 0; i < names.size() && (Eigen::Index)i < model.size(); ++i) {
                std::stringstream ss;
                ss << names[i] << " = " << model[(Eigen::Index)i];
                disp.emplace_back(ss.str());
            }
            return disp;
        }
        const Eigen::Index tail = model.size() / 2;
        Eigen::Index idx = 0;
        for (size_t i = 0; i < names.size() && idx < tail; ++i)
            for (size_t j = i + 1; j < names.size() && idx < tail; ++j, ++idx) {
                std::stringstream ss;
                ss << names[i] << " = " << model[idx] << '*' << names[j] << " + " << model[tail + idx];
                disp.emplace_back(ss.str());
            }
        return disp;
    };
}

Transformation Transformation::Standard(const std::vector<std::string> &columnNames)
{
    Transformation t;
    t.names = columnNames;
    t.Displayer = make_displayer(columnNames, false);
    t.Name = "Standard transformation";
    return t;
}

Transformation Transformation::StandardMedianEstimator(const std::vector<std::string> &columnNames)
{
    Transformation t;
    t.Stepper = vsom::MedianStepper{};
    t.names = columnNames;
    t.Displayer = make_displayer(columnNames, false);
    t.Name = "Standard median estimator transformation";
    return t;
}

Transformation Transformation::CombinatorialLinearRegression(const std::vector<std::string> &columnNames)
{
    Transformation t;
    t.Comparer = vsom::ClrComparer{};
    t.Stepper = vsom::ClrStepper{};
    t.Length = vsom::ClrLength{};
    t.names = columnNames;
    t.Displayer = make_displayer(columnNames, true);
    t.Name = "Linear regression";
    return t;
}

int Transformation::kind() const noexcept
{
    const bool sc = Comparer.target<vsom::StandardComparer>() != nullptr;
    const bool ss = Stepper.target<vsom::StandardStepper>() != nullptr;
    const bool ms = Stepper.target<vsom::MedianStepper>() != nullptr;
    const bool il = Length.target<vsom::IdentityLength>() != nullptr;
    if (sc && ss && il)
        return vsom::Standard;
    if (sc && ms && il)
        return vsom::Median;
    if (Comparer.target<vsom::ClrComparer>() && Stepper.target<vsom::ClrStepper>() && Length.target<vsom::ClrLength>())
        return vsom::Clr;
    return vsom::Custom;
}

// ---------------------------------------------------------------------------------------------
// SomIndex (reference src/SomIndex.cpp:10-44)
// ---------------------------------------------------------------------------------------------
SomIndex::SomIndex(size_t inX, size_t inY) noexcept : x{inX}, y{inY} {}
SomIndex::SomIndex(const Som &map, size_t index) noexcept
    : x{index % map.getWidth()}, y{(index - index % map.getWidth()) / map.getHeight()} {}   // divides by HEIGHT (Q10)
size_t SomIndex::getSomIndex(const Som &map) { return map.getWidth() * y + x; }
size_t SomIndex::getX() const noexcept { return x; }
size_t SomIndex::getY() const noexcept { return y; }
void SomIndex::setX(size_t ix) noexcept { x = ix; }
void SomIndex::setY(size_t iy) noexcept { y = iy; }

// ---------------------------------------------------------------------------------------------
// ArrayDataLoader / DataSet (reference src/DataSet.cpp:8-176)
// ---------------------------------------------------------------------------------------------
ArrayDataLoader::ArrayDataLoader(const float *rows, size_t nrows, size_t depth, std::optional<size_t> maxLoadCount)
    : IDataLoader(maxLoadCount), m_rows(rows, rows + nrows * depth), m_nrows(nrows), m_depth(depth),
      m_weights(depth, 1.0f), m_binary(depth, 0), m_continuous(depth, 1)
{
    for (size_t i = 0; i < depth; ++i)
        m_names.push_back("c" + std::to_string(i));
}

size_t ArrayDataLoader::load()
{
    const size_t chunk = m_maxLoadCount.value_or(m_nrows);
    const size_t end = std::min(m_currentIndex + chunk, m_nrows);
    data.clear();
    data.reserve(end - m_currentIndex);
    for (size_t r = m_currentIndex; r < end; ++r) {
        RowData row;
        row.values = Eigen::VectorXf((Eigen::Index)m_depth);
        std::memcpy(row.values.data(), &m_rows[r * m_depth], m_depth * sizeof(float));
        row.valid.assign(m_depth, 1);
        data.push_back(std::move(row));
    }
    const size_t n = end - m_currentIndex;
    m_currentIndex = end >= m_nrows ? 0 : end;   // wrap: the stream is back at its start
    return n;
}

// rows straight into the caller's (pinned) buffer: one copy instead of a heap object per row and two copies
bool ArrayDataLoader::peekFlat(size_t &rows)
{
    const size_t chunk = m_maxLoadCount.value_or(m_nrows);
    rows = std::min(m_currentIndex + chunk, m_nrows) - m_currentIndex;
    return true;
}

size_t ArrayDataLoader::loadFlat(float *dst, size_t rows)
{
    const size_t chunk = m_maxLoadCount.value_or(m_nrows);
    const size_t end = std::min(m_currentIndex + chunk, m_nrows);
    const size_t n = std::min(rows, end - m_currentIndex);
    data.clear();
    const float *src = &m_rows[m_currentIndex * m_depth];
    const size_t total = n * m_depth;
    // a 12.8 MB chunk is ~1 ms of one core's memcpy: a few threads bring it well under the device step it hides behind
    const size_t nthreads = total >= (1u << 20) ? std::min<size_t>(4, std::max(1u, std::thread::hardware_concurrency())) : 1;
    if (nthreads <= 1) {
        std::memcpy(dst, src, total * sizeof(float));
    } else {
        std::vector<std::thread> pool;
        const size_t per = (total + nthreads - 1) / nthreads;
        for (size_t t = 0; t < nthreads; ++t) {
            const size_t a = t * per, b = std::min(total, a + per);
            if (a < b)
                pool.emplace_back([=] { std::memcpy(dst + a, src + a, (b - a) * sizeof(float)); });
        }
        for (auto &th : pool)
            th.join();
    }
    m_currentIndex = end >= m_nrows ? 0 : end;   // wrap: the stream is back at its start
    return n;
}

std::vector<RowData> ArrayDataLoader::getPreview(size_t count)
{
    std::vector<RowData> out;
    for (size_t r = 0; r < std::min(count, m_nrows); ++r) {
        RowData row;
        row.values = Eigen::VectorXf((Eigen::Index)m_depth);
        for (size_t d = 0; d < m_depth; ++d)
            row.values[(Eigen::Index)d] = m_rows[r * m_depth + d];
        row.valid.assign(m_depth, 1);
        out.push_back(std::move(row));
    }
    return out;
}

void ArrayDataLoader::setColumnSpec(const std::vector<ColumnSpec> columnSpec) noexcept
{
    for (size_t i = 0; i < columnSpec.size() && i < m_depth; ++i) {
        m_names[i] = columnSpec[i].name;
        m_weights[i] = columnSpec[i].weight;
        m_binary[i] = columnSpec[i].isBinary ? 1 : 0;
        m_continuous[i] = columnSpec[i].isBinary ? 0 : 1;
    }
}

const std::vector<ColumnSpec> ArrayDataLoader::getColumnSpec() noexcept
{
    std::vector<ColumnSpec> out;
    for (size_t i = 0; i < m_depth; ++i)
        out.emplace_back(m_names[i], m_weights[i], m_binary[i]);
    return out;
}

// The per-row containers of the reference (vector<VectorXf> data, vector<vector<int>> valid, the
// DataRow views) are built on first use: the training drivers only need the contiguous staging copy,
// and materialising three heap objects per row costs more host time per chunk than the device step.
void DataSet::ensureRows() const
{
    if (m_rowsBuilt)
        return;
    data.clear();
    valid.clear();
    allData.clear();
    data.reserve(n);
    valid.reserve(n);
    allData.reserve(n);
    size_t cur = 0;
    if (m_fromFlat) {             // the chunk came through IDataLoader::loadFlat: rows from the staging copy, all valid
        const float *flat = m_flat[m_cur].p;
        for (; cur < n; ++cur) {
            Eigen::VectorXf v((Eigen::Index)depth);
            std::memcpy(v.data(), flat + cur * depth, depth * sizeof(float));
            data.push_back(std::move(v));
            valid.emplace_back(depth, 1);
            allData.push_back(DataRow{&data.back(), &valid.back(), &lastBMU[cur]});
        }
        m_rowsBuilt = true;
        return;
    }
    for (auto &row : _loader.data) {
        if (cur >= n)
            break;
        data.push_back(row.values);
        valid.push_back(row.valid);
        allData.push_back(DataRow{&data.back(), &valid.back(), &lastBMU[cur]});
        ++cur;
    }
    m_rowsBuilt = true;
}

const std::vector<DataSet::DataRow> DataSet::getAll() const
{
    ensureRows();
    return allData;
}
std::vector<DataSet::DataRow> DataSet::getAll()
{
    ensureRows();
    return allData;
}

std::vector<Eigen::VectorXf> DataSet::getPreviewData(size_t count) const
{
    std::vector<Eigen::VectorXf> out;
    for (auto &item : _loader.getPreview(count))
        out.emplace_back(item.values);
    return out;
}

Eigen::VectorXf DataSet::getData(size_t i) const
{
    ensureRows();
    if ((m_fromFlat ? n : _loader.data.size()) > i && data.size() > i)
        return data[i];
    return Eigen::VectorXf::Zero((Eigen::Index)_loader.getDepth());
}

const Eigen::VectorXi DataSet::getValidity(size_t i) const
{
    ensureRows();
    if (n > i && valid.size() > i) {      // a view of the row's own flags, whatever their count (DataSet.cpp:44-45): a loader may
        Eigen::VectorXi v((Eigen::Index)valid[i].size());   // hand over rows longer than its depth (IDX images that are not 28 x 28)
        for (size_t d = 0; d < valid[i].size(); ++d)
            v[(Eigen::Index)d] = valid[i][d];
        return v;
    }
    return Eigen::VectorXi::Zero((Eigen::Index)_loader.getDepth());
}

static Eigen::ArrayXi to_arrayxi(const std::vector<int> &src)
{
    Eigen::ArrayXi a((Eigen::Index)src.size());
    for (size_t i = 0; i < src.size(); ++i)
        a[(Eigen::Index)i] = src[i];
    return a;
}

const Eigen::ArrayXi DataSet::getBinary() const { return to_arrayxi(_loader.getBinary()); }
const Eigen::ArrayXi DataSet::getContinuous() const { return to_arrayxi(_loader.getContinuous()); }

const Eigen::VectorXf DataSet::getWeights() const
{
    const auto w = _loader.getWeights();
    Eigen::VectorXf v((Eigen::Index)w.size());
    for (size_t i = 0; i < w.size(); ++i)
        v[(Eigen::Index)i] = w[i];
    return v;
}

float DataSet::getWeight(size_t i) { return _loader.getWeight(i); }
const std::vector<std::string> DataSet::getNames() const noexcept { return _loader.getNames(); }
std::string DataSet::getName(size_t i) const { return _loader.getName(i); }
const std::vector<size_t> &DataSet::getLastBMU() const noexcept { return lastBMU; }
size_t &DataSet::getLastBMU(size_t i)
{
    assert(n > i);
    return lastBMU[i];
}
size_t DataSet::size() const { return n; }

void DataSet::addVector(Eigen::VectorXf v)
{
    if ((size_t)v.rows() == _loader.getDepth()) {
        ensureRows();
        depth = _loader.getDepth();
        m_flat[m_cur].reserve((data.size() + 1) * depth);
        for (size_t d = 0; d < depth; ++d)
            m_flat[m_cur].p[data.size() * depth + d] = v[(Eigen::Index)d];
        data.push_back(v);
        n += 1;
    } else {
        std::cout << "Added vector size does not correspond to data set depth!\n";
    }
}

void DataSet::resetStreamLoadPosition() noexcept { loadedNumberOfChunks = 0; }
bool DataSet::hasReadWholeDataStream() const noexcept { return loadedNumberOfChunks > 0 && _loader.isAtStartOfDataStream(); }

DataSet::Pinned::~Pinned()
{
    if (p && pinned)
        (void)vsom_host_free(p);
    else
        std::free(p);
}

void DataSet::Pinned::reserve(size_t nfloats)
{
    if (nfloats <= cap)
        return;
    const size_t want = std::max(nfloats, cap * 2);
    // pinned when a device is present (asynchronous copies); plain pageable memory otherwise, so the
    // loaders / DataSet stay usable for inspection on a machine without a GPU (training still needs one)
    void *q = nullptr;
    bool qpinned = true;
    if (vsom_host_alloc(&q, want * sizeof(float)) != 0 || !q) {
        q = std::malloc(want * sizeof(float));
        qpinned = false;
        if (!q)
            throw std::bad_alloc();
    }
    if (p) {
        std::copy(p, p + cap, static_cast<float *>(q));
        if (pinned)
            (void)vsom_host_free(p);
        else
            std::free(p);
    }
    p = static_cast<float *>(q);
    pinned = qpinned;
    cap = want;
}

// the reference passes DataSet by value (SOM.hpp:78-82); the copy owns its row views and staging
DataSet::DataSet(const DataSet &o)
    : data{o.data}, valid{o.valid}, index{o.index}, lastBMU{o.lastBMU}, _loader{o._loader}, depth{o.depth}, n{o.n},
      loadedNumberOfChunks{o.loadedNumberOfChunks}, _verbose{o._verbose}, m_rowsBuilt{o.m_rowsBuilt}, m_fromFlat{o.m_fromFlat}
{
    if (m_rowsBuilt) {
        allData.reserve(o.allData.size());
        for (size_t k = 0; k < o.allData.size() && k < data.size(); ++k)
            allData.push_back(DataRow{&data[k], &valid[k], &lastBMU[k]});
    }
    const size_t nf = std::max(n, data.size()) * depth;
    if (nf && o.m_flat[o.m_cur].p) {
        m_flat[0].reserve(nf);
        std::copy(o.m_flat[o.m_cur].p, o.m_flat[o.m_cur].p + nf, m_flat[0].p);
    }
}

// DataSet.cpp:118-160: reload, rebuild the row views, lastBMU := 0 (:136-137)
void DataSet::loadNextDataFromStream()
{
    if (_loader.isAtStartOfDataStream())
        loadedNumberOfChunks = 0;
    // the other pinned buffer: an asynchronous copy of the previous chunk may still read the current one
    m_cur ^= 1;
    depth = _loader.getDepth();
    size_t numberOfRows = 0, peek = 0;
    if (_loader.peekFlat(peek)) {
        // the loader writes the chunk straight into the pinned staging buffer (no per-row heap objects, one copy);
        // the reference's per-row containers are still built on first use (ensureRows)
        m_flat[m_cur].reserve(peek * depth);
        numberOfRows = _loader.loadFlat(m_flat[m_cur].p, peek);
        m_fromFlat = true;
    } else {
        numberOfRows = _loader.load();
        depth = _loader.getDepth();
        m_fromFlat = false;
        m_flat[m_cur].reserve(numberOfRows * depth);
        float *flat = m_flat[m_cur].p;
        size_t cur = 0;
        for (auto &row : _loader.data) {
            if (cur >= numberOfRows)
                break;
            const size_t have = std::min<size_t>((size_t)row.values.size(), depth);
            std::memcpy(flat + cur * depth, row.values.data(), have * sizeof(float));
            for (size_t d = have; d < depth; ++d)
                flat[cur * depth + d] = 0.f;
            ++cur;
        }
    }
    n = numberOfRows;
    lastBMU.assign(numberOfRows, 0);
    m_rowsBuilt = false;          // data / valid / allData are rebuilt on first use
    index.resize(numberOfRows);
    for (size_t k = 0; k < index.size(); ++k)
        index[k] = k;
    shuffle();
    ++loadedNumberOfChunks;
    if (_verbose)
        std::cout << "Loaded " << loadedNumberOfChunks << " number of chunks\n";
}

void DataSet::display() const
{
    std::cout << "Number of samples: " << (m_fromFlat ? n : _loader.data.size()) << "\nVector length: " << _loader.getDepth() << "\n";
}
size_t DataSet::vectorLength() const { return _loader.getDepth(); }
// the reference shuffles an index vector that nothing reads (DataSet.cpp:143-146,173-176): order = load order
void DataSet::shuffle() {}

// ---------------------------------------------------------------------------------------------
// Som
// ---------------------------------------------------------------------------------------------
static int g_default_device = 0;
void Som::setDefaultDevice(int device) { g_default_device = device; }

// Devices a Som trains on.  Nothing named: the default device, i.e. the single-GPU path.  Several GPUs are
// OPT-IN: VSOM_DEVICES="0,1,2,3" (or Som::setDevices) names them, and more than one entry makes the batch-map
// training members run through the vsom_group_* entry points (phase 1 sample-sharded, phase 2 node-sharded,
// all-gathers over xGMI).  A list that repeats a device rehearses that flow on one GPU (the library then
// copies between the members instead of calling RCCL, include/vsom_hip.h).  Rounds 1-2 grouped every visible
// GPU by default; the RCCL transport of a group has not run on N > 1 real devices yet,
// so a caller who never asked for several GPUs no longer gets it.
static std::vector<int> g_devices;
static bool g_devices_set = false;
void Som::setDevices(const std::vector<int> &devices)
{
    g_devices = devices;
    g_devices_set = true;
}

// Arithmetic of the Standard update chains for the Soms created from now on (include/vsom_hip.h,
// vsom_update_mode): strict (default: everything bit-identical to the reference); sigma-contracted
// (VSOM_UPDATE_FMA_SIGMA: map / BMUs / MSE bit-identical over whole schedules, sigmaMap within 1e-5, a sixth
// fewer instructions); contracted (VSOM_UPDATE_FMA: one-epoch tolerance only -- a schedule drifts off the
// reference's trajectory).  VSOM_UPDATE_MODE=strict|sigma|contracted in the environment selects it without a
// code change.
static int g_update_mode = -1;
void Som::setContractedArithmetic(bool on) { g_update_mode = on ? VSOM_UPDATE_FMA : VSOM_UPDATE_STRICT; }
void Som::setUpdateArithmetic(int mode)
{
    if (mode != VSOM_UPDATE_STRICT && mode != VSOM_UPDATE_FMA && mode != VSOM_UPDATE_FMA_SIGMA)
        throw std::invalid_argument("Som::setUpdateArithmetic: unknown vsom_update_mode");
    g_update_mode = mode;
}
static int update_mode()
{
    if (g_update_mode >= 0)
        return g_update_mode;
    const char *e = std::getenv("VSOM_UPDATE_MODE");
    const std::string m = e ? e : "";
    if (m == "contracted" || m == "fma")
        return VSOM_UPDATE_FMA;
    if (m == "sigma" || m == "fma_sigma" || m == "sigma_contracted")
        return VSOM_UPDATE_FMA_SIGMA;
    return VSOM_UPDATE_STRICT;
}

static std::vector<int> training_devices(size_t nodes)
{
    if (g_devices_set)
        return g_devices.empty() ? std::vector<int>{g_default_device} : g_devices;
    std::vector<int> out;
    if (const char *e = std::getenv("VSOM_DEVICES")) {
        std::stringstream ss(e);
        std::string tok;
        while (std::getline(ss, tok, ','))
            if (!tok.empty())
                out.push_back(std::atoi(tok.c_str()));
        if (!out.empty())
            return out;
    }
    (void)nodes;
    return {g_default_device};   // nothing named: one GPU (several are opt-in, see above)
}

static void check(int rc, const char *what)
{
    if (rc != 0)
        throw std::runtime_error(std::string(what) + ": " + vsom_last_error());
}

void Som::createContext()
{
    const int kind = transform.kind();
    if (kind == vsom::Custom) {
        ctx = nullptr;   // state-less shell: accessors work on zeros, training throws
        return;
    }
    const std::vector<int> devs = training_devices(width * height);
    if (devs.size() > 1) {
        check(vsom_group_create(&grp, (int)devs.size(), devs.data(), (uint32_t)width, (uint32_t)height, (uint32_t)inLen, kind),
              "vsom_group_create");
        ctx = vsom_group_ctx(grp, 0);     // searches, getters and the online path use member 0 (the state is replicated)
        replicasStale = false;
        check(vsom_group_set_update_mode(grp, update_mode()), "vsom_group_set_update_mode");
        return;
    }
    check(vsom_create(&ctx, devs[0], (uint32_t)width, (uint32_t)height, (uint32_t)inLen, kind), "vsom_create");
    check(vsom_set_update_mode(ctx, update_mode()), "vsom_set_update_mode");
}

void Som::destroyContext()
{
    if (grp)
        vsom_group_destroy(grp);   // owns its members, ctx included
    else if (ctx)
        vsom_destroy(ctx);
    grp = nullptr;
    ctx = nullptr;
}

// the online members (trainSingle / trainBasicSom) train member 0 only -- that path is strictly sequential
// in samples, "replicas only" -- so before the next sharded epoch the other members take over its state
void Som::syncReplicas()
{
    if (!grp || !replicasStale)
        return;
    const size_t N = width * height;
    std::vector<float> m(N * depth), s(N * depth), S(N * depth), w(N);
    std::vector<uint64_t> hh(N);
    check(vsom_get_state(ctx, m.data(), s.data(), S.data(), w.data(), hh.data()), "vsom_get_state");
    check(vsom_group_set_state(grp, m.data(), s.data(), S.data(), w.data(), hh.data()), "vsom_group_set_state");
    replicasStale = false;
}

// Under a group the sigmaMap / weightMap rows of the other shards are gathered on a second stream and only
// joined before the next phase 2 (csrc/vsom_group.hip, deferred gathers): every read of member 0's sigma or
// weight between asynchronous epochs -- U-matrix, raw distances, state getters -- joins them first.
void Som::joinGroup() const
{
    if (grp)
        check(vsom_group_synchronize(grp), "vsom_group_synchronize");
}

void Som::requireDevicePath(const char *what) const
{
    if (!ctx)
        throw std::runtime_error(std::string(what) +
                                 ": this Transformation is not one of Standard / StandardMedianEstimator / "
                                 "CombinatorialLinearRegression; with custom std::function hooks only the training "
                                 "path (distance, findBmu, findLocalBmu, train*) runs, on the host");
}

void Som::Construct(size_t inWidth, size_t inHeight, size_t inDepth, std::vector<std::string> names)
{
    width = inWidth;
    height = inHeight;
    depth = inDepth;
    uMatrix.assign(inWidth * inHeight, 0.0);
    transform.names = names;
    createContext();   // Som.cpp:11-48: all state zero (vsom_create zero-fills)
    hostStale = true;
}

static size_t in_len_from_depth(const Transformation &t, size_t depth)
{
    if (t.kind() != vsom::Clr)
        return depth;
    // depth = J(J-1) (Transformation.cpp:162-165); the depth constructor is used this way by
    // tests/performance/perf_tests.cpp:338-339
    size_t J = (size_t)std::llround((1.0 + std::sqrt(1.0 + 4.0 * (double)depth)) / 2.0);
    if (J * (J - 1) != depth)
        throw std::invalid_argument("depth is not J*(J-1) for any J (CombinatorialLinearRegression)");
    return J;
}

Som::Som(size_t w, size_t h, DataSet dataset, Transformation transformation) : transform{transformation}, _isTraining{false}
{
    inLen = dataset.vectorLength();
    // the reference passes the *input* length as depth here (SOM.hpp:81); the model length follows
    // from the transformation
    Construct(w, h, transform.Length(inLen), dataset.getNames());
}

Som::Som(size_t w, size_t h, size_t d, Transformation transformation) : transform{transformation}, _isTraining{false}
{
    inLen = in_len_from_depth(transform, d);
    Construct(w, h, d, std::vector<std::string>{});
}

// Som.cpp:51-83: size the map from the file (getSizeFromFile), zero state, default (Standard)
// transformation, then load.  A binary checkpoint of this build carries its own dimensions.
Som::Som(const char *filename) : _isTraining{false}
{
    width = height = depth = 0;
    {
        std::ifstream f(filename, std::ios::binary);
        uint64_t magic = 0;
        f.read((char *)&magic, sizeof(magic));
        if (!f || magic != 0x314d4f5356ull) {
            size_t w = 0, h = 0, d = 0;
            if (!vsom::read_octave_dims(filename, w, h, d)) {
                std::cout << "Could not open file " << filename << " for reading. Quitting...\n";
                std::exit(EXIT_FAILURE);   // :1306-1310
            }
            width = w;
            height = h;
            depth = d;
            inLen = d;
            uMatrix.assign(width * height, 0.0);
            createContext();
        }
    }
    load(filename);
}

Som::Som(const Som &som)
    : transform{som.transform}, metrics{}, uMatrix{som.uMatrix}, _isTraining{false}, height{som.height},
      width{som.width}, depth{som.depth}, inLen{som.inLen}
{
    createContext();
    copyStateFrom(som);
    _isTraining.store(som._isTraining.load());
}

// the reference's implicit copy takes the whole model (SOM.hpp:56-61); here the state lives on the device, or
// -- custom std::function hooks, ctx == nullptr -- in the host arrays: getState / setState handle both
void Som::copyStateFrom(const Som &other)
{
    // (one side device-resident and the other host-resident is fine too: the copy goes through host arrays)
    const size_t N = width * height;
    std::vector<float> m(N * depth), s(N * depth), S(N * depth), w(N);
    std::vector<uint64_t> hh(N);
    other.getState(m.data(), s.data(), S.data(), w.data(), hh.data());
    setState(m.data(), s.data(), S.data(), w.data(), hh.data());
    if (!ctx)
        hostStale = false;       // the host arrays ARE the state: nothing to refresh from
}

Som &Som::operator=(const Som &other)
{
    if (this == &other)
        return *this;
    destroyContext();
    transform = other.transform;
    uMatrix = other.uMatrix;
    height = other.height;
    width = other.width;
    depth = other.depth;
    inLen = other.inLen;
    createContext();
    hostStale = true;
    hMap.clear(); hSigma.clear(); hS.clear(); hWeight.clear(); hHits.clear();
    copyStateFrom(other);
    _isTraining.store(other._isTraining.load());
    return *this;
}

Som::~Som() { destroyContext(); }

void Som::setState(const float *map, const float *sigma, const float *S, const float *weight, const uint64_t *hits)
{
    if (!ctx) {   // custom hooks: the host arrays ARE the state
        hostEnsure();
        const size_t N = width * height;
        if (map) hMap.assign(map, map + N * depth);
        if (sigma) hSigma.assign(sigma, sigma + N * depth);
        if (S) hS.assign(S, S + N * depth);
        if (weight) hWeight.assign(weight, weight + N);
        if (hits) hHits.assign(hits, hits + N);
        return;
    }
    if (grp)
        check(vsom_group_set_state(grp, map, sigma, S, weight, hits), "vsom_group_set_state");
    else
        check(vsom_set_state(ctx, map, sigma, S, weight, hits), "vsom_set_state");
    hostStale = true;
}

void Som::getState(float *map, float *sigma, float *S, float *weight, uint64_t *hits) const
{
    if (!ctx) {
        hostEnsure();
        if (map) std::copy(hMap.begin(), hMap.end(), map);
        if (sigma) std::copy(hSigma.begin(), hSigma.end(), sigma);
        if (S) std::copy(hS.begin(), hS.end(), S);
        if (weight) std::copy(hWeight.begin(), hWeight.end(), weight);
        if (hits) std::copy(hHits.begin(), hHits.end(), hits);
        return;
    }
    joinGroup();
    check(vsom_get_state(ctx, map, sigma, S, weight, hits), "vsom_get_state");
}

void Som::refreshHost() const
{
    if (!hostStale)
        return;
    const size_t N = width * height;
    hMap.assign(N * depth, 0.f);
    hSigma.assign(N * depth, 0.f);
    hWeight.assign(N, 0.f);
    hHits.assign(N, 0);
    if (ctx) {
        joinGroup();
        check(vsom_get_state(ctx, hMap.data(), hSigma.data(), nullptr, hWeight.data(), hHits.data()), "vsom_get_state");
    }
    hostStale = false;
}

// ---- accessors ---------------------------------------------------------------------------------
UMatrix Som::getUMatrix() const noexcept { return UMatrix{uMatrix, width, height}; }
size_t Som::getHeight() const noexcept { return height; }
size_t Som::getWidth() const noexcept { return width; }
size_t Som::getDepth() const noexcept { return depth; }
size_t Som::getIndex(SomIndex i) const noexcept { return i.getY() * width + i.getX(); }

static Eigen::VectorXf row_of(const std::vector<float> &a, size_t row, size_t depth)
{
    Eigen::VectorXf v((Eigen::Index)depth);
    for (size_t d = 0; d < depth; ++d)
        v[(Eigen::Index)d] = a[row * depth + d];
    return v;
}

Eigen::VectorXf Som::getWeigthMap() const noexcept
{
    refreshHost();
    Eigen::VectorXf v((Eigen::Index)hWeight.size());
    for (size_t i = 0; i < hWeight.size(); ++i)
        v[(Eigen::Index)i] = hWeight[i];
    return v;
}

std::vector<size_t> Som::getBmuHits() const noexcept
{
    refreshHost();
    return std::vector<size_t>(hHits.begin(), hHits.end());
}

Eigen::VectorXf Som::getNeuron(SomIndex i) const noexcept { return getNeuron(getIndex(i)); }
Eigen::VectorXf Som::getNeuron(size_t i) const noexcept
{
    refreshHost();
    return row_of(hMap, i, depth);
}
Eigen::VectorXf Som::getSigmaNeuron(SomIndex i) const noexcept { return getSigmaNeuron(getIndex(i)); }
Eigen::VectorXf Som::getSigmaNeuron(size_t i) const noexcept
{
    refreshHost();
    return row_of(hSigma, i, depth);
}
std::vector<std::string> Som::getNeuronStrings(SomIndex index) const noexcept { return transform.Displayer(getNeuron(index)); }
std::vector<std::string> Som::getSigmaNeuronStrings(SomIndex index) const noexcept { return transform.Displayer(getSigmaNeuron(index)); }

static float extreme(const std::vector<float> &a, size_t depth, size_t col, bool want_max)
{
    assert(col < depth);
    float best = a[col];
    for (size_t r = 0; r * depth < a.size(); ++r) {
        float v = a[r * depth + col];
        if (want_max ? best < v : v < best)
            best = v;
    }
    return best;
}

float Som::getMaxValueOfFeature(size_t c) const { refreshHost(); return extreme(hMap, depth, c, true); }
float Som::getMinValueOfFeature(size_t c) const { refreshHost(); return extreme(hMap, depth, c, false); }
float Som::getMaxSigmaOfFeature(size_t c) const { refreshHost(); return extreme(hSigma, depth, c, true); }
float Som::getMinSigmaOfFeature(size_t c) const { refreshHost(); return extreme(hSigma, depth, c, false); }
Som::Metrics Som::getMetrics() const noexcept { return metrics; }
bool Som::isTraining() const noexcept { return _isTraining; }
bool Som::isCompatibleWithData(DataSet &data) const noexcept { return transform.Length(data.vectorLength()) == depth; }

void Som::display() const
{
    refreshHost();
    std::cout << "Map size: " << width * height << "\nMap width: " << width << "\nMap height: " << height
              << "\nM size: " << depth << "\n\n";
    for (size_t i = 0; i < height; ++i) {
        for (size_t j = 0; j < width; ++j)
            std::cout << hHits[i * width + j] << "\t";
        std::cout << "\n";
    }
}

void Som::displayUMatrix() const
{
    std::cout << "\nU-matrix:\n[";
    for (size_t i = 0; i < height; i++) {
        for (size_t j = 0; j < width; j++)
            std::cout << uMatrix[i * width + j] << " ";
        std::cout << "; ";
    }
    std::cout << "]\n";
}

// Som.cpp:977-997: glibc srand/rand, node-major, dim-minor; everything else zero
void Som::randomInitialize(int seed, float sigma)
{
    std::srand((unsigned)seed);
    metrics = Metrics{depth};
    const size_t N = width * height;
    std::vector<float> m(N * depth), z(N * depth, 0.f), w(N, 0.f);
    std::vector<uint64_t> hh(N, 0);
    for (size_t i = 0; i < N; ++i)
        for (size_t d = 0; d < depth; ++d)
            m[i * depth + d] = (static_cast<float>(std::rand() % static_cast<int>((2000 * sigma))) - (1000.f * sigma)) / 1000.f;
    std::fill(uMatrix.begin(), uMatrix.end(), 0.0);
    setState(m.data(), z.data(), z.data(), w.data(), hh.data());
}

void Som::addBmu(SomIndex pos)   // Som.cpp:1189-1192
{
    if (!ctx) {
        hostEnsure();
        hHits[getIndex(pos)] += 1;
        return;
    }
    const size_t N = width * height;
    std::vector<uint64_t> hh(N);
    check(vsom_get_state(ctx, nullptr, nullptr, nullptr, nullptr, hh.data()), "vsom_get_state");
    hh[getIndex(pos)] += 1;
    if (grp && !replicasStale)
        check(vsom_group_set_state(grp, nullptr, nullptr, nullptr, nullptr, hh.data()), "vsom_group_set_state");
    else
        check(vsom_set_state(ctx, nullptr, nullptr, nullptr, nullptr, hh.data()), "vsom_set_state");
    hostStale = true;
}

double Som::calculateNeighbourhoodWeight(const size_t &currentX, const size_t &currentY, const size_t &bmuX,
                                         const size_t &bmuY, const double &currentSigma)
{
    return vsom_neighbourhood_weight(currentX, currentY, bmuX, bmuY, currentSigma);
}

// ---- search ------------------------------------------------------------------------------------
void Som::stageOne(const Eigen::VectorXf &v) const
{
    requireDevicePath("search");
    if ((size_t)v.size() != inLen)
        throw std::invalid_argument("sample length does not match the map");
    check(vsom_upload_chunk(ctx, v.data(), 1), "vsom_upload_chunk");
}

SomIndex Som::findBmu(const Eigen::VectorXf &v) const
{
    Eigen::VectorXf ones = Eigen::VectorXf::Ones(v.size());
    return findBmu(v, ones, ones);
}

SomIndex Som::findBmu(const Eigen::VectorXf &v, const Eigen::VectorXf &valid, const Eigen::VectorXf &weights) const
{
    if (!ctx) {   // custom hooks: host search (vsom_custom.cpp)
        const size_t idx = hostFindBmu(v, valid, weights);
        return SomIndex(idx % width, idx / width);
    }
    if ((size_t)v.size() != inLen)
        throw std::invalid_argument("sample length does not match the map");
    uint64_t idx = 0;
    check(vsom_find_bmu(ctx, v.data(), &idx, nullptr), "vsom_find_bmu");
    return SomIndex((size_t)idx % width, (size_t)idx / width);   // Som.cpp:306
}

SomIndex Som::findLocalBmu(const Eigen::VectorXf &v, const Eigen::VectorXf &valid, const size_t &lastBMUref,
                           const Eigen::VectorXf &weights) const
{
    if (!ctx) {
        const size_t idx = hostFindLocalBmu(v, valid, lastBMUref, weights);
        return SomIndex(idx % width, idx / width);
    }
    if ((size_t)v.size() != inLen)
        throw std::invalid_argument("sample length does not match the map");
    uint64_t idx = lastBMUref;
    check(vsom_find_local_bmu(ctx, v.data(), lastBMUref, &idx, nullptr), "vsom_find_local_bmu");
    return SomIndex((size_t)idx % width, (size_t)idx / width);
}

double Som::euclidianWeightedDist(const SomIndex &pos, const Eigen::VectorXf &v, const Eigen::VectorXf &valid,
                                  const Eigen::VectorXf &weights) const
{
    return euclidianWeightedDist(pos.getY() * width + pos.getX(), v, valid, weights);
}

double Som::euclidianWeightedDist(const size_t &pos, const Eigen::VectorXf &v, const Eigen::VectorXf &valid,
                                  const Eigen::VectorXf &weights) const
{
    if (!ctx)
        return hostDist(pos, v, valid, weights);
    if ((size_t)v.size() != inLen)
        throw std::invalid_argument("sample length does not match the map");
    float d = 0.f;
    check(vsom_dist_single(ctx, v.data(), pos, &d), "vsom_dist_single");
    return (double)d;
}

// ---- batch training ----------------------------------------------------------------------------
float Som::trainBatchSomEpoch(DataSet &dataset, double currentSigma, bool isFirst)
{
    if (!ctx)
        return hostBatchEpoch(dataset, currentSigma, isFirst);
    const size_t B = dataset.size();
    // B == 0 is not skipped: the reference's epoch over an empty chunk still rewrites every neuron
    // (zero model vector, NaN sigma, zero weight -- Som.cpp:840-875), e.g. after the zero-row load that
    // ends every chunked MnistDataLoader pass
    std::vector<uint64_t> lb(B);
    if (!isFirst)
        for (size_t s = 0; s < B; ++s)
            lb[s] = dataset.getLastBMU(s);
    float mse = 0.f;
    if (grp) {   // several GPUs: samples sharded in phase 1, nodes in phase 2 (include/vsom_hip.h, vsom_group_*)
        syncReplicas();
        check(vsom_group_upload_chunk(grp, dataset.contiguous(), B), "vsom_group_upload_chunk");
        if (!isFirst)
            check(vsom_group_set_last_bmu(grp, lb.data()), "vsom_group_set_last_bmu");
        check(vsom_group_batch_epoch(grp, currentSigma, isFirst ? 1 : 0, &mse), "vsom_group_batch_epoch");
        check(vsom_group_get_last_bmu(grp, lb.data()), "vsom_group_get_last_bmu");
    } else {
        check(vsom_upload_chunk(ctx, dataset.contiguous(), B), "vsom_upload_chunk");
        if (!isFirst)
            check(vsom_set_last_bmu(ctx, lb.data()), "vsom_set_last_bmu");
        check(vsom_batch_epoch(ctx, currentSigma, isFirst ? 1 : 0, &mse), "vsom_batch_epoch");
        check(vsom_get_last_bmu(ctx, lb.data()), "vsom_get_last_bmu");
    }
    for (size_t s = 0; s < B; ++s)
        dataset.getLastBMU(s) = (size_t)lb[s];   // *data.lastBMU = index (Som.cpp:777,800)
    hostStale = true;
    return mse;
}

// Som.cpp:716-754.  Same control flow; the chunk loop is software-pipelined: while the device trains
// on chunk k (asynchronous epoch), the host loads chunk k+1 from the loader and starts its
// host->device copy (vsom_prefetch_chunk from the DataSet's pinned staging buffer).  The reference
// reloads every chunk every epoch (:737), so this overlap recurs for the whole run.
void Som::trainBatchSom(DataSet &data, size_t numberOfEpochs, double sigma0, double sigmaDecay,
                        bool updateUMatrixAfterEpoch)
{
    if (!ctx)
        return hostTrainBatchSom(data, numberOfEpochs, sigma0, sigmaDecay, updateUMatrixAfterEpoch);
    {   // Som.cpp:719 resets the metrics WITHOUT the mutex (a thread copying them under it races with the reassignment);
        // here the reset takes it, like every later write
        const std::lock_guard<std::mutex> lock(metricsMutex);
        metrics = Som::Metrics(numberOfEpochs);
    }
    auto sigmaOf = [&](size_t e) { return sigma0 * std::exp(-sigmaDecay * static_cast<double>(e)); };   // :727
    bool have = false;                // a loaded chunk is waiting in `data`, its copy is in flight
    size_t B = 0;
    // one GPU: the context's entry points; several: the group's (same contracts, include/vsom_hip.h)
    syncReplicas();
    auto prefetch = [&](const float *x, size_t n) {
        check(grp ? vsom_group_prefetch_chunk(grp, x, n) : vsom_prefetch_chunk(ctx, x, n), "vsom_prefetch_chunk");
    };
    auto commit = [&] { check(grp ? vsom_group_commit_chunk(grp) : vsom_commit_chunk(ctx), "vsom_commit_chunk"); };
    auto epochAsync = [&](double sg, int first) {
        check(grp ? vsom_group_batch_epoch_async(grp, sg, first) : vsom_batch_epoch_async(ctx, sg, first), "vsom_batch_epoch_async");
    };
    auto lastBmu = [&](uint64_t *out) {
        check(grp ? vsom_group_get_last_bmu(grp, out) : vsom_get_last_bmu(ctx, out), "vsom_get_last_bmu");
    };
    auto getMse = [&](float *out) { check(grp ? vsom_group_get_mse(grp, out) : vsom_get_mse(ctx, out), "vsom_get_mse"); };
    // leaving (normally or by exception): the members' deferred sigmaMap / weightMap gathers are joined
    struct Join {
        vsom_group *g;
        ~Join()
        {
            if (g)
                (void)vsom_group_synchronize(g);
        }
    } join{grp};
    for (size_t i = 0; i < numberOfEpochs; ++i) {
        std::cout << "Training VSOM epoch " << i << "/" << numberOfEpochs << '\n';
        const auto sigma = sigmaOf(i);
        if (sigma < 1.0)
            return;   // :729-730
        auto meanSquareError = float{0.0f};
        auto countDataChunks = size_t{0};
        if (!have && !data.hasReadWholeDataStream()) {   // :735 (the first chunk may already be in flight)
            data.loadNextDataFromStream();
            B = data.size();
            prefetch(data.contiguous(), B);
            have = true;
        }
        // the next epoch will run (and therefore reload the stream from its start, :737,749)?
        const bool another = i + 1 < numberOfEpochs && sigmaOf(i + 1) >= 1.0;
        // set only where the next epoch's first chunk really was loaded and its copy started: an epoch
        // that found the stream already read to its end (a caller who loaded the single chunk before
        // train()) processes no chunk, records 0/0 = NaN and resets the stream like Som.cpp:735-749
        bool prefetchedNext = false;
        while (have) {
            const size_t Bcur = B;
            // (also for an empty chunk: the reference's epoch then zeroes the map, see trainBatchSomEpoch)
            commit();   // lastBMU := 0 (DataSet.cpp:136-137)
            epochAsync(sigma, i == 0 ? 1 : 0);
            have = false;
            const bool last = data.hasReadWholeDataStream();
            if (!last) {
                data.loadNextDataFromStream();   // host work beside the device epoch
                B = data.size();
                prefetch(data.contiguous(), B);
                have = true;
            } else if (another) {
                // last chunk of this epoch: load the next epoch's first chunk beside it (the BMUs of
                // this chunk would be wiped by that reload in the reference too, DataSet.cpp:136-137)
                data.resetStreamLoadPosition();   // :749
                data.loadNextDataFromStream();
                B = data.size();
                prefetch(data.contiguous(), B);
                prefetchedNext = true;
            } else if (Bcur > 0) {
                // the chunk still held by `data` when training ends keeps its BMUs (Som.cpp:777,800)
                std::vector<uint64_t> lb(Bcur);
                lastBmu(lb.data());
                for (size_t s = 0; s < Bcur; ++s)
                    data.getLastBMU(s) = (size_t)lb[s];
            }
            float mse = 0.f;
            getMse(&mse);
            meanSquareError += mse;
            ++countDataChunks;
            if (last) {
                have = false;
                break;
            }
        }
        hostStale = true;
        meanSquareError /= static_cast<float>(countDataChunks);   // :743
        {
            const std::lock_guard<std::mutex> lock(metricsMutex);
            metrics.MeanSquaredError[i] = meanSquareError;
        }
        if (prefetchedNext)
            have = true;              // chunk 0 of epoch i+1 is already loaded and on its way
        else
            data.resetStreamLoadPosition();   // :749
        if (updateUMatrixAfterEpoch)
            updateUMatrix(data.getWeights());   // :751-752
    }
}

// ---- online training ---------------------------------------------------------------------------
static int decay_code(Som::WeigthDecayFunction f)
{
    return f == Som::WeigthDecayFunction::Exponential ? VSOM_EXPONENTIAL
           : f == Som::WeigthDecayFunction::InverseProportional ? VSOM_INVERSE_PROPORTIONAL : VSOM_BATCHMAP;
}

Som::TrainingReturnValue Som::trainSingle(const Eigen::VectorXf &v, const Eigen::VectorXf &valid, const Eigen::VectorXf &weights,
                                          const double eta, const double sigma, size_t &lastBMU,
                                          const WeigthDecayFunction weightDecayFunction)
{
    if (!ctx)
        return hostTrainSingle(v, valid, weights, eta, sigma, lastBMU, weightDecayFunction);
    if ((size_t)v.size() != inLen)
        throw std::invalid_argument("sample length does not match the map");
    Eigen::VectorXf residual((Eigen::Index)vsom_residual_len(ctx));
    uint64_t lb = lastBMU, bmu = 0;
    float dist = 0.f;
    check(vsom_train_single(ctx, v.data(), eta, sigma, &lb, decay_code(weightDecayFunction), residual.data(), &dist, &bmu),
          "vsom_train_single");
    lastBMU = (size_t)lb;
    hostStale = true;
    replicasStale = grp != nullptr;
    return TrainingReturnValue{SomIndex((size_t)bmu % width, (size_t)bmu / width), residual, dist};
}

void Som::trainBasicSom(DataSet &data, size_t numberOfEpochs, double eta0, double etaDecay, double sigma0,
                        double sigmaDecay, WeigthDecayFunction weightDecayFunction, bool updateUMatrixAfterEpoch)
{
    if (!ctx)
        return hostTrainBasicSom(data, numberOfEpochs, eta0, etaDecay, sigma0, sigmaDecay, weightDecayFunction, updateUMatrixAfterEpoch);
    if (grp) {
        check(vsom_group_synchronize(grp), "vsom_group_synchronize");
        replicasStale = true;                 // the online path trains member 0 only (sequential in samples)
    }
    {   // Som.cpp:1139, under the mutex here (see trainBatchSom)
        const std::lock_guard<std::mutex> lock(metricsMutex);
        metrics = Som::Metrics(numberOfEpochs);
    }
    for (size_t i = 0; i < numberOfEpochs; ++i) {
        auto eta = eta0 * std::exp(-etaDecay * static_cast<double>(i));         // :1145
        auto sigma = sigma0 * std::exp(-sigmaDecay * static_cast<double>(i));   // :1146
        if (sigma < 1.0)
            sigma = 1.0;   // :1148-1149
        std::cout << "Epoch: " << i + 1 << "/" << numberOfEpochs << "\teta: " << eta << "\tsigma: " << sigma << "\n";
        float meanSquareError{0.0};
        size_t countDataChunks{0};
        std::vector<uint64_t> lbScratch;
        // same software pipeline as trainBatchSom: chunk k+1 is loaded and copied while the device
        // walks the B sequential trainSingle steps of chunk k (:1161-1171)
        bool have = false, staged = false;
        size_t B = 0;
        if (!data.hasReadWholeDataStream()) {
            data.loadNextDataFromStream();
            B = data.size();
            // the epoch's first chunk: the previous epoch has been waited for, nothing runs that a prefetch on the copy
            // stream could overlap with -- copy and staging go straight onto the context's stream, no events, no wait (the
            // DataSet's pinned buffer is not rewritten before the chunk after next is loaded)
            check(vsom_upload_chunk_async(ctx, data.contiguous(), B), "vsom_upload_chunk_async");
            have = true;
            staged = true;
        }
        while (have) {
            const size_t Bcur = B;
            // meanSquareError is ONE running float over all chunks of the epoch (:1153,1167): the device
            // keeps it across chunks (first chunk starts at 0); an empty chunk adds nothing
            if (!staged)
                check(vsom_commit_chunk(ctx), "vsom_commit_chunk");
            staged = false;
            have = false;
            const bool last = data.hasReadWholeDataStream();
            if (last && Bcur > 0) {
                // the epoch's last chunk: the sample loop, its lastBMU (:895,1163) and the running MSE in ONE call and one
                // synchronisation (the reference's own 20-row scenario is such a chunk every epoch)
                lbScratch.resize(Bcur);
                check(vsom_train_online_chunk_fetch(ctx, eta, sigma, decay_code(weightDecayFunction), countDataChunks == 0 ? 1 : 0,
                                                    lbScratch.data(), &meanSquareError),
                      "vsom_train_online_chunk_fetch");
                for (size_t s = 0; s < Bcur; ++s)
                    data.getLastBMU(s) = (size_t)lbScratch[s];
            } else {
                check(vsom_train_online_chunk_acc(ctx, eta, sigma, decay_code(weightDecayFunction), countDataChunks == 0 ? 1 : 0,
                                                  nullptr),
                      "vsom_train_online_chunk_acc");
                if (!last) {
                    data.loadNextDataFromStream();
                    B = data.size();
                    check(vsom_prefetch_chunk(ctx, data.contiguous(), B), "vsom_prefetch_chunk");
                    have = true;
                }
                check(vsom_get_mse(ctx, &meanSquareError), "vsom_get_mse");   // the running value so far
            }
            ++countDataChunks;
        }
        meanSquareError /= static_cast<float>(countDataChunks);   // :1175
        {
            const std::lock_guard<std::mutex> lock(metricsMutex);
            metrics.MeanSquaredError[i] = meanSquareError;
        }
        data.resetStreamLoadPosition();
        hostStale = true;
        if (updateUMatrixAfterEpoch)
            updateUMatrix(data.getWeights());   // :1183-1184
    }
    std::cout << "\rTraining SOM:100%\n";
}

// Som.cpp:1113-1132: dispatch; exceptions are printed, not propagated
void Som::train(DataSet &data, size_t numberOfEpochs, double eta0, double etaDecay, double sigma0, double sigmaDecay,
                WeigthDecayFunction weightDecayFunction, bool updateUMatrixAfterEpoch)
{
    _isTraining = true;
    try {
        switch (weightDecayFunction) {
        case WeigthDecayFunction::BatchMap:
            trainBatchSom(data, numberOfEpochs, sigma0, sigmaDecay, updateUMatrixAfterEpoch);
            break;
        default:
            trainBasicSom(data, numberOfEpochs, eta0, etaDecay, sigma0, sigmaDecay, weightDecayFunction, updateUMatrixAfterEpoch);
        }
    } catch (const std::exception &e) {
        std::cerr << e.what() << '\n';
    }
    _isTraining = false;
}

// ---- consumers of the search outside the training loop --------------------
// The distance work (restricted BMU search, all-node distances, raw sigma-normalised distances)
// runs on the device; the scalar post-processing follows the reference line by line on the host.

// Som.cpp:313-332
SomIndex Som::findRestrictedBmu(const Eigen::VectorXf &v, const Eigen::VectorXf &valid, const size_t minBmuHits,
                                const Eigen::VectorXf &weights) const
{
    if (!ctx) {   // caller's hooks: host search (vsom_custom.cpp)
        const size_t idx = hostFindRestrictedBmu(v, valid, minBmuHits, weights);
        return SomIndex(idx % width, idx / width);
    }
    if ((size_t)v.size() != inLen)
        throw std::invalid_argument("sample length does not match the map");
    uint64_t idx = 0;
    check(vsom_find_restricted_bmu(ctx, v.data(), minBmuHits, &idx, nullptr), "vsom_find_restricted_bmu");
    return SomIndex((size_t)idx % width, (size_t)idx / width);
}

// Som.cpp:457-487
std::vector<double> Som::findRestrictedBmd(const Eigen::VectorXf &v, const Eigen::VectorXf &valid, size_t minBmuHits,
                                           const Eigen::VectorXf &weights) const
{
    const size_t N = width * height;
    std::vector<float> d(N);
    if (!ctx) {   // caller's hooks: the distances through Comparer on the host
        hostEnsure();
        for (size_t i = 0; i < N; ++i)
            d[i] = hHits[i] >= minBmuHits ? (float)hostDist(i, v, valid, weights) : 0.f;
    } else {
        if ((size_t)v.size() != inLen)
            throw std::invalid_argument("sample length does not match the map");
        refreshHost();
        check(vsom_distances_single(ctx, v.data(), d.data()), "vsom_distances_single");
    }
    std::vector<double> dist(N, -1);
    double C = 0;
    for (size_t i = 0; i < N; ++i) {
        if (hHits[i] >= minBmuHits) {
            dist[i] = (double)d[i];
            dist[i] = (double)std::exp(-dist[i] * dist[i] / 2);   // :473
            C += dist[i];
        } else
            dist[i] = 0;
    }
    for (size_t i = 0; i < N; ++i)
        dist[i] /= C;
    return dist;
}

// Som.cpp:143-157 (the built-in use passes valid = weights = 1; other masks are not supported here)
double Som::euclidianWeightedDistRaw(const size_t &pos, const Eigen::VectorXf &v, const Eigen::VectorXf &valid,
                                     const Eigen::VectorXf &weights) const
{
    if (!ctx)
        return hostDistRaw(pos, v, valid, weights);
    joinGroup();
    stageOne(v);
    uint64_t node = pos, row = 0;
    float d = 0.f;
    check(vsom_distances_raw(ctx, &node, &row, 1, 0, &d), "vsom_distances_raw");
    return (double)d;
}

// Som.cpp:999-1111: mean sigma-normalised distance to the 3/5/8 neighbours, diagonals weighted 0.3
void Som::updateUMatrix(const Eigen::VectorXf &)
{
    if (ctx)
        joinGroup();   // raw_dist_kernel reads sigma of every node: the deferred sigmaMap gather must have landed
    // neighbour offsets in the order the reference adds them for an interior node (:1017-1024):
    // W, E, S(i+1), N(i-1), NW(i-1,j-1), SW(i+1,j-1), NE(i-1,j+1), SE(i+1,j+1)
    static const int DI[8] = {0, 0, 1, -1, -1, 1, -1, 1};
    static const int DJ[8] = {-1, 1, 0, 0, -1, -1, 1, 1};
    const size_t N = width * height;
    std::vector<uint64_t> nodes, nbrs;
    std::vector<int> slot;   // slot[node*8 + k] = index into the pair list or -1
    slot.assign(N * 8, -1);
    for (size_t i = 0; i < height; ++i)
        for (size_t j = 0; j < width; ++j)
            for (int k = 0; k < 8; ++k) {
                long ni = (long)i + DI[k], nj = (long)j + DJ[k];
                if (ni < 0 || nj < 0 || ni >= (long)height || nj >= (long)width)
                    continue;
                slot[(i * width + j) * 8 + k] = (int)nodes.size();
                nodes.push_back(i * width + j);
                nbrs.push_back((size_t)ni * width + (size_t)nj);
            }
    std::vector<float> d(nodes.size());
    if (!ctx) {   // caller's hooks: euclidianWeightedDistRaw involves none (Som.cpp:143-157), host arithmetic
        hostEnsure();
        const Eigen::VectorXf ones = Eigen::VectorXf::Constant((Eigen::Index)depth, 1.f);   // :1002-1003
        for (size_t k = 0; k < nodes.size(); ++k)
            d[k] = (float)hostDistRaw((size_t)nodes[k], row_of(hMap, (size_t)nbrs[k], depth), ones, ones);
    } else if (!nodes.empty())
        check(vsom_distances_raw(ctx, nodes.data(), nbrs.data(), nodes.size(), 1, d.data()), "vsom_distances_raw");
    const double diagonalFactor = 0.3;
    auto R = [&](size_t n, int k) { return (double)d[(size_t)slot[n * 8 + k]]; };
    enum { Wk = 0, Ek = 1, Sk = 2, Nk = 3, NWk = 4, SWk = 5, NEk = 6, SEk = 7 };
    for (size_t i = 0; i < height; ++i)
        for (size_t j = 0; j < width; ++j) {
            const size_t n = i * width + j;
            double u;
            if (j > 0 && i > 0 && j < (width - 1) && i < (height - 1))
                u = (R(n, Wk) + R(n, Ek) + R(n, Sk) + R(n, Nk) + R(n, NWk) * diagonalFactor + R(n, SWk) * diagonalFactor +
                     R(n, NEk) * diagonalFactor + R(n, SEk) * diagonalFactor) / 8;
            else if (i == 0 && j > 0 && j < (width - 1))
                u = (R(n, Wk) + R(n, Ek) + R(n, Sk) + R(n, SWk) * diagonalFactor + R(n, SEk) * diagonalFactor) / 5;
            else if (i == (height - 1) && j > 0 && j < (width - 1))
                u = (R(n, Wk) + R(n, Ek) + R(n, Nk) + R(n, NWk) * diagonalFactor + R(n, NEk) * diagonalFactor) / 5;
            else if (j == 0 && i > 0 && i < (height - 1))
                u = (R(n, Ek) + R(n, Sk) + R(n, Nk) + R(n, NEk) * diagonalFactor + R(n, SEk) * diagonalFactor) / 5;
            else if (j == (width - 1) && i > 0 && i < (height - 1))
                u = (R(n, Wk) + R(n, Sk) + R(n, Nk) + R(n, NWk) * diagonalFactor + R(n, SWk) * diagonalFactor) / 5;
            else if (j == 0 && i == 0)
                u = (R(n, Ek) + R(n, Sk) + R(n, SEk) * diagonalFactor) / 3;
            else if (j == (width - 1) && i == 0)
                u = (R(n, Wk) + R(n, Sk) + R(n, SWk) * diagonalFactor) / 3;
            else if (j == 0 && i == (height - 1))
                u = (R(n, Ek) + R(n, Nk) + R(n, NEk) * diagonalFactor) / 3;
            else if (j == (width - 1) && i == (height - 1))
                u = (R(n, Wk) + R(n, Nk) + R(n, NWk) * diagonalFactor) / 3;
            else
                u = 0;
            uMatrix[n] = u;
        }
}

// Som.cpp:490-523.  The reference's `ones` array is constructed but never filled (ArrayXf(rows,1),
// :495); it is taken to be all ones here.  log is libm's logf (Eigen's vectorised plog may differ
// in the last bits).
double Som::evaluate(const DataSet &data) const
{
    const size_t n = data.size();
    if (n == 0)
        return 0.0;
    std::vector<uint64_t> bmu(n);
    std::vector<float> dist(n);
    const Eigen::ArrayXi continuous = data.getContinuous(), binary = data.getBinary();
    if (!ctx) {   // caller's hooks: findBmu / euclidianWeightedDist per sample with val = validity * continuous (:503-505)
        hostEnsure();
        const Eigen::VectorXf w = data.getWeights();
        for (size_t i = 0; i < n; ++i) {
            const Eigen::VectorXi validity = data.getValidity(i);
            Eigen::VectorXf val((Eigen::Index)validity.size());
            for (Eigen::Index d = 0; d < validity.size(); ++d)
                val[d] = (float)(validity[d] * continuous[d]);
            const Eigen::VectorXf x = data.getData(i);
            bmu[i] = hostFindBmu(x, val, w);
            dist[i] = (float)hostDist((size_t)bmu[i], x, val, w);
        }
    } else {
        refreshHost();
        check(vsom_upload_chunk(ctx, data.contiguous(), n), "vsom_upload_chunk");
        check(vsom_bmu_batch(ctx, bmu.data(), dist.data()), "vsom_bmu_batch");   // findBmu + euclidianWeightedDist(bmu)
    }
    double error = 0;
    for (size_t i = 0; i < n; i++) {
        const Eigen::VectorXi validity = data.getValidity(i);
        const Eigen::VectorXf x = data.getData(i);
        double bsum = 0;   // binaryError.dot(binaryError)
        const size_t dlen = std::min<size_t>((size_t)x.size(), depth);
        float acc = 0.f;
        for (size_t d = 0; d < dlen; ++d) {
            const float val = (float)(validity[(Eigen::Index)d] * continuous[(Eigen::Index)d]);
            const float m = hMap[(size_t)bmu[i] * depth + d];
            float be = std::log(m) * x[(Eigen::Index)d] + std::log(1.0f - m) * (1.0f - x[(Eigen::Index)d]);
            if (std::isnan(be) || std::isinf(be))
                be = -99999.f;                                             // :512
            be = be * (float)binary[(Eigen::Index)d] * val;                // :514
            acc += be * be;
        }
        bsum = (double)acc;
        error += 1. / (static_cast<double>(i) + 1.0) * ((double)dist[i] + std::sqrt(bsum) - error);   // :519
    }
    return error;
}

// Som.cpp:631-714
int Som::measureSimilarity(const DataSet *data, int numOfSigmas, size_t minBmuHits) const
{
    const size_t n = data->size();
    if (n == 0)
        return true;
    std::vector<uint64_t> bmus(n);
    if (!ctx) {   // caller's hooks: findRestrictedBmu per sample on the host (:652)
        hostEnsure();
        const Eigen::VectorXf w = data->getWeights();
        for (size_t i = 0; i < n; ++i) {
            const Eigen::VectorXi validity = data->getValidity(i);
            Eigen::VectorXf val((Eigen::Index)validity.size());
            for (Eigen::Index d = 0; d < validity.size(); ++d)
                val[d] = (float)validity[d];
            bmus[i] = hostFindRestrictedBmu(data->getData(i), val, minBmuHits, w);
        }
    } else {
        refreshHost();
        check(vsom_upload_chunk(ctx, data->contiguous(), n), "vsom_upload_chunk");
        check(vsom_bmu_restricted_batch(ctx, minBmuHits, bmus.data(), nullptr), "vsom_bmu_restricted_batch");
    }
    bool success = true;
    float maxValue{-99999999.f};
    size_t maxValueDataSetRow{0};
    bool last{false};
    for (size_t i = 0; i < n + 1; i++) {
        if (i == n) {
            i = maxValueDataSetRow;
            last = true;
        }
        const Eigen::VectorXf v = data->getData(i);
        const size_t pos = (size_t)bmus[i];
        const Eigen::VectorXi validEigen = data->getValidity(i);
        for (Eigen::Index d = 0; d < v.size() && (size_t)d < depth; d++) {
            const float sg = hSigma[pos * depth + (size_t)d], m = hMap[pos * depth + (size_t)d];
            const float sM = sg > 0.00001f ? 0.00001f : sg;                 // :658 (select as written)
            float delta = (v[d] - m) / sM / (float)numOfSigmas;            // :671
            const float mn = m - sM * (float)numOfSigmas, mx = m + sM * (float)numOfSigmas;   // :675-677
            if (delta > maxValue) {
                maxValue = static_cast<float>(std::fabs(delta));
                maxValueDataSetRow = i;
            }
            if (validEigen[d] && last)
                if ((v[d] < mn || v[d] > mx) && validEigen[d])
                    success = false;
        }
        if (last)
            break;
    }
    return success;
}

// Som.cpp:525-566: draws a model vector from the restricted distribution of each sample; returns the
// draw of the LAST sample (as the reference does).  Non-deterministic (std::random_device).
size_t Som::variationalAutoEncoder(const DataSet *data, size_t minBmuHits) const
{
    std::random_device rd;
    std::mt19937 gen(rd());
    size_t modelVector{0};
    for (size_t i = 0; i < data->size(); ++i) {
        Eigen::VectorXf v = data->getData(i);
        Eigen::VectorXf val = data->getValidity(i).cast<float>();
        auto probability = findRestrictedBmd(v, val, minBmuHits, data->getWeights());
        std::discrete_distribution<size_t> d(probability.begin(), probability.end());
        modelVector = d(gen);
    }
    return modelVector;
}

// Som.cpp:568-623: prints a logit-approximated normal sample per feature around a drawn model vector
int Som::autoEncoder(const DataSet *data, size_t minBmuHits) const
{
    std::srand((unsigned)(time(NULL) + clock()));
    for (size_t i = 0; i < data->size(); i++) {
        Eigen::VectorXf v = data->getData(i);
        auto bmuInt = variationalAutoEncoder(data, minBmuHits);
        SomIndex bmu(bmuInt % width, bmuInt / width);
        const Eigen::VectorXf sg = getSigmaNeuron(bmu), m = getNeuron(bmu);
        for (Eigen::Index k = 0; k < v.size(); k++) {
            std::cout << v(k) << "\n";
            double L = (double)(std::rand() % (int)(1000)) / 1000;
            double Nn = std::log(L / (1 - L)) / 1.6 * sg[k] + m[k];
            std::cout << data->getName((size_t)k) << "\t" << Nn << "\t\n";
        }
        std::cout << "\n";
    }
    return true;
}

// ---- checkpoint: lossless little-endian binary of this build (the reference's Octave text format,
//      Som.cpp:1209-1597, is a "next" row) --------------------------------------------------------------
// Som.cpp:1209-1294: the Octave/Matlab text checkpoint, byte-compatible with the reference's writer
// (vsom_checkpoint.cpp).  Like the reference's it keeps six decimals and no SMap; saveBinary below
// is this build's lossless alternative.
void Som::save(const char *fileName) const
{
    vsom::Checkpoint c;
    c.width = width;
    c.height = height;
    c.depth = depth;
    c.resize();
    getState(c.map.data(), c.sigma.data(), nullptr, c.weight.data(), c.hits.data());
    c.U = uMatrix;
    c.U.resize(width * height, 0.0);
    if (!vsom::write_octave(fileName, c)) {
        std::cout << "Could not open file " << fileName << " for writing. Quitting...\n";
        std::exit(EXIT_FAILURE);   // Som.cpp:1213-1217
    }
}

// [MI355X build] lossless binary checkpoint (all five state arrays incl. SMap, the sample length
// and the transformation kind); Som::load recognises it by its magic.
void Som::saveBinary(const char *fileName) const
{
    const size_t N = width * height;
    std::vector<float> m(N * depth), s(N * depth), S(N * depth), w(N);
    std::vector<uint64_t> hh(N);
    getState(m.data(), s.data(), S.data(), w.data(), hh.data());
    std::ofstream f(fileName, std::ios::binary);
    if (!f) {
        std::cout << "Could not open file " << fileName << " for writing. Quitting...\n";
        std::exit(EXIT_FAILURE);
    }
    const uint64_t hdr[6] = {0x314d4f5356ull /* "VSOM1" */, width, height, depth, inLen, (uint64_t)transform.kind()};
    f.write((const char *)hdr, sizeof(hdr));
    f.write((const char *)m.data(), m.size() * 4);
    f.write((const char *)s.data(), s.size() * 4);
    f.write((const char *)S.data(), S.size() * 4);
    f.write((const char *)w.data(), w.size() * 4);
    f.write((const char *)hh.data(), hh.size() * 8);
}

// Som.cpp:1296-1341.  The reference assigns `height` from "# rows:" and `width` from "# columns:" and
// returns a vector whose length is the last number of the line after "# ndims:" (length 1 otherwise);
// a file that cannot be opened ends the process (:1305-1309).
Eigen::VectorXf Som::getSizeFromFile(const char *fileName)
{
    size_t w = width, h = height, d = 1;
    if (!vsom::read_octave_dims(fileName, w, h, d)) {
        std::cout << "Could not open file " << fileName << " for reading. Quitting...\n";
        std::exit(EXIT_FAILURE);
    }
    if (w != width || h != height) {
        width = w;
        height = h;
        uMatrix.assign(width * height, 0.0);
        destroyContext();
        createContext();
        hostStale = true;
    }
    return Eigen::VectorXf((Eigen::Index)d);
}

// Som.cpp:1343-1597: reads the Octave text checkpoint into the EXISTING map (dimensions come from
// the constructor, Som.cpp:51-83, or from the object as it is); SMap is not part of that format and
// is left as it is.  A file that starts with this build's binary magic is loaded losslessly instead.
void Som::load(const char *fileName)
{
    std::ifstream f(fileName, std::ios::binary);
    if (!f) {
        std::cout << "Could not open file " << fileName << " for reading. Quitting...\n";
        std::exit(EXIT_FAILURE);   // :1359-1363
    }
    uint64_t hdr[6] = {0, 0, 0, 0, 0, 0};
    f.read((char *)hdr, sizeof(hdr));
    if (f && hdr[0] == 0x314d4f5356ull) {
        destroyContext();
        width = hdr[1];
        height = hdr[2];
        depth = hdr[3];
        inLen = hdr[4];
        transform = hdr[5] == vsom::Median ? Transformation::StandardMedianEstimator({})
                    : hdr[5] == vsom::Clr  ? Transformation::CombinatorialLinearRegression({})
                                           : Transformation::Standard({});
        uMatrix.assign(width * height, 0.0);
        createContext();
        const size_t N = width * height;
        std::vector<float> m(N * depth), s(N * depth), S(N * depth), w(N);
        std::vector<uint64_t> hh(N);
        f.read((char *)m.data(), m.size() * 4);
        f.read((char *)s.data(), s.size() * 4);
        f.read((char *)S.data(), S.size() * 4);
        f.read((char *)w.data(), w.size() * 4);
        f.read((char *)hh.data(), hh.size() * 8);
        if (!f)
            throw std::runtime_error("truncated VSOM1 checkpoint");
        setState(m.data(), s.data(), S.data(), w.data(), hh.data());
        return;
    }
    f.close();
    vsom::Checkpoint c;
    c.width = width;
    c.height = height;
    c.depth = depth;
    c.resize();
    getState(c.map.data(), c.sigma.data(), nullptr, c.weight.data(), c.hits.data());   // sections absent from the file keep their values
    c.U = uMatrix;
    c.U.resize(width * height, 0.0);
    if (!vsom::read_octave(fileName, c)) {
        std::cout << "Could not open file " << fileName << " for reading. Quitting...\n";
        std::exit(EXIT_FAILURE);
    }
    if (c.width * c.height != width * height)
        throw std::runtime_error("checkpoint does not match the map size");
    width = c.width;      // "# rows" / "# columns" of the last 2-D section (:1470-1483)
    height = c.height;
    uMatrix = c.U;
    setState(c.map.data(), c.sigma.data(), nullptr, c.weight.data(), c.hits.data());
}
