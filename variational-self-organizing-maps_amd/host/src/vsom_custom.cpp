// vsom_custom.cpp -- host execution of the training path for a Transformation whose Comparer / Stepper are
// the CALLER's std::functions (Transformation::kind() == vsom::Custom; tests/test1.cpp:56-84 builds one).
//
// User callables cannot run on the GPU, so a Som constructed with them keeps its state in host arrays
// (row-major N x depth, the mirror's hMap / hSigma / hS / hWeight / hHits) and its training members run
// here, calling the hooks exactly where the reference does: distance Som.cpp:124-141, findBmu :291-309,
// findLocalBmu :335-454, batch epoch :756-879, online step :885-947, drivers :716-754 / :1135-1187.
// This is the interface's escape hatch, not a fallback of the device path: the three built-in
// transformations never come here (they have no host implementation of the epoch at all), and nothing in
// this file is reached from libvsom_hip.so.
//
// fp32 operation order follows the reference's Eigen expressions with every scalar*vector product
// narrowed to float first; r.dot(r) is summed in Eigen 3.4's packet order (below), so a custom hook that
// computes what a built-in does reproduces the device path's (and the oracle's) bits.
#include "vsom_api.hpp"
#include "vsom_hip.h"

#include <cmath>
#include <iostream>
#include <stdexcept>

namespace {

using Vec = Eigen::VectorXf;

// r.dot(r) as Eigen 3.4 evaluates it with SSE packets and no FMA: lanes k = 0..3 of two packet
// accumulators over blocks of 8, the second folded into the first, one more whole packet if there is
// one, lanes combined as (l0 + l2) + (l1 + l3), the tail added element by element; fewer than 4 elements
// are summed in order.
float packet_order_square_sum(const Vec &r)
{
    const size_t n = (size_t)r.size();
    if (n == 0)
        return 0.f;
    auto sq = [&](size_t i) { const float x = r[(Eigen::Index)i]; return x * x; };
    const size_t whole = n & ~size_t{3}, pairs = n & ~size_t{7};
    if (whole == 0) {
        float s = sq(0);
        for (size_t i = 1; i < n; ++i)
            s = s + sq(i);
        return s;
    }
    float lane[2][4];
    for (size_t k = 0; k < 4; ++k)
        lane[0][k] = sq(k);
    if (whole > 4) {
        for (size_t k = 0; k < 4; ++k)
            lane[1][k] = sq(4 + k);
        for (size_t base = 8; base < pairs; base += 8)
            for (size_t half = 0; half < 2; ++half)
                for (size_t k = 0; k < 4; ++k)
                    lane[half][k] = lane[half][k] + sq(base + 4 * half + k);
        for (size_t k = 0; k < 4; ++k)
            lane[0][k] = lane[0][k] + lane[1][k];
        if (whole > pairs)
            for (size_t k = 0; k < 4; ++k)
                lane[0][k] = lane[0][k] + sq(pairs + k);
    }
    float s = (lane[0][0] + lane[0][2]) + (lane[0][1] + lane[0][3]);
    for (size_t i = whole; i < n; ++i)
        s = s + sq(i);
    return s;
}

// a.dot(b) in the same packet order (Som.cpp:156: two different vectors)
float packet_order_dot(const Vec &a, const Vec &b)
{
    const size_t n = (size_t)a.size();
    if (n == 0)
        return 0.f;
    auto pr = [&](size_t i) { return a[(Eigen::Index)i] * b[(Eigen::Index)i]; };
    const size_t whole = n & ~size_t{3}, pairs = n & ~size_t{7};
    if (whole == 0) {
        float s = pr(0);
        for (size_t i = 1; i < n; ++i)
            s = s + pr(i);
        return s;
    }
    float lane[2][4];
    for (size_t k = 0; k < 4; ++k)
        lane[0][k] = pr(k);
    if (whole > 4) {
        for (size_t k = 0; k < 4; ++k)
            lane[1][k] = pr(4 + k);
        for (size_t base = 8; base < pairs; base += 8)
            for (size_t half = 0; half < 2; ++half)
                for (size_t k = 0; k < 4; ++k)
                    lane[half][k] = lane[half][k] + pr(base + 4 * half + k);
        for (size_t k = 0; k < 4; ++k)
            lane[0][k] = lane[0][k] + lane[1][k];
        if (whole > pairs)
            for (size_t k = 0; k < 4; ++k)
                lane[0][k] = lane[0][k] + pr(pairs + k);
    }
    float s = (lane[0][0] + lane[0][2]) + (lane[0][1] + lane[0][3]);
    for (size_t i = whole; i < n; ++i)
        s = s + pr(i);
    return s;
}

Vec row(const std::vector<float> &a, size_t node, size_t depth)
{
    Vec v((Eigen::Index)depth);
    for (size_t d = 0; d < depth; ++d)
        v[(Eigen::Index)d] = a[node * depth + d];
    return v;
}

void requireLength(const Vec &v, size_t depth, const char *hook)
{
    if ((size_t)v.size() != depth)
        throw std::runtime_error(std::string("custom Transformation::") + hook + " returned " + std::to_string(v.size()) +
                                 " values for a model vector of " + std::to_string(depth));
}

Vec asFloat(const Eigen::VectorXi &v)
{
    Vec out((Eigen::Index)v.size());
    for (Eigen::Index i = 0; i < v.size(); ++i)
        out[i] = (float)v[i];
    return out;
}

}   // namespace

void Som::hostEnsure() const
{
    refreshHost();                                   // zero state of Construct on first use
    if (hS.size() != hMap.size())
        hS.assign(hMap.size(), 0.f);
}

// Som.cpp:124-141
double Som::hostDist(size_t pos, const Vec &v, const Vec &valid, const Vec &weights) const
{
    hostEnsure();
    Vec floorSigma((Eigen::Index)depth);             // :131  select(sigma < 1e-5, 1e-5, sigma)
    for (size_t d = 0; d < depth; ++d) {
        const float s = hSigma[pos * depth + d];
        floorSigma[(Eigen::Index)d] = s < 0.00001f ? 0.00001f : s;
    }
    Vec validWeights((Eigen::Index)valid.size());    // :134
    for (Eigen::Index d = 0; d < valid.size(); ++d)
        validWeights[d] = valid[d] * weights[d];
    const Vec r = transform.Comparer(v, row(hMap, pos, depth), floorSigma, validWeights);   // :136
    return (double)packet_order_square_sum(r);       // :140
}

// Som.cpp:291-309: node 0 first, strict '<', lowest index wins ties, NaN never wins
size_t Som::hostFindBmu(const Vec &v, const Vec &valid, const Vec &weights) const
{
    size_t best = 0;
    double bestDist = hostDist(0, v, valid, weights);
    for (size_t i = 0; i < width * height; ++i) {
        const double d = hostDist(i, v, valid, weights);
        if (d < bestDist) {
            bestDist = d;
            best = i;
        }
    }
    return best;
}

// Som.cpp:335-454.  The coordinate arithmetic is size_t like the reference's: "-1" offsets wrap and the
// min() against width-1 / height-1 then clamps them to the far border; moving in Y evaluates nothing
// (its loop counter starts at SIZE_MAX), moving in X evaluates the three nodes ahead.
size_t Som::hostFindLocalBmu(const Vec &v, const Vec &valid, size_t start, const Vec &weights) const
{
    static const size_t stepX[8] = {~size_t{0}, 0, 1, 1, 1, 0, ~size_t{0}, ~size_t{0}};
    static const size_t stepY[8] = {1, 1, 1, 0, ~size_t{0}, ~size_t{0}, ~size_t{0}, 0};
    auto clampTo = [](size_t value, size_t last) { return value < last ? value : last; };
    size_t anchor = start, probe = start, best = start;
    double bestDist = hostDist(start, v, valid, weights);
    auto consider = [&](size_t x, size_t y) {
        const size_t node = y * width + x;
        const double d = hostDist(node, v, valid, weights);
        if (d < bestDist) {
            bestDist = d;
            best = node;
        }
    };
    for (;;) {
        const size_t px = probe % width, py = probe / width, ax = anchor % width;
        if (probe == anchor) {                       // first round: the 8 neighbours (:362-385)
            for (int i = 0; i < 8; ++i)
                consider(clampTo(px + stepX[i], width - 1), clampTo(py + stepY[i], height - 1));
            if (best == anchor)
                return best;
            probe = best;
        } else {
            if (px - ax) {                           // :390-403
                const size_t x = clampTo(px + px - ax, width - 1);
                for (int i = -1; i < 2; ++i)
                    consider(x, clampTo(py + (size_t)(long long)i, height - 1));
            }
            if (best == probe)                       // :440-443
                return best;
            anchor = probe;
            probe = best;
        }
    }
}

// Som.cpp:756-879
float Som::hostBatchEpoch(DataSet &dataset, double currentSigma, bool isFirst)
{
    hostEnsure();
    const size_t B = dataset.size(), N = width * height;
    const Vec weights = dataset.getWeights();
    float mse = 0.f;                                 // the reference's atomic<float>: summed in sample order here
    std::vector<Vec> samples, valids;
    samples.reserve(B);
    valids.reserve(B);
    for (size_t j = 0; j < B; ++j) {                 // phase 1 (:762-806)
        samples.push_back(dataset.getData(j));
        valids.push_back(asFloat(dataset.getValidity(j)));
        size_t &last = dataset.getLastBMU(j);
        const size_t idx = isFirst ? hostFindBmu(samples[j], valids[j], weights) : hostFindLocalBmu(samples[j], valids[j], last, weights);
        last = idx;
        hHits[idx] += 1;
        const Vec residual = transform.Comparer(samples[j], row(hMap, idx, depth), row(hS, idx, depth), valids[j]);
        mse += packet_order_square_sum(residual) / static_cast<float>(B);
    }
    std::vector<float> newMap(N * depth), newSigma(N * depth);
    for (size_t node = 0; node < N; ++node) {        // phase 2 (:809-876)
        const SomIndex here(*this, node);            // y divides by HEIGHT (SomIndex.cpp:13-18)
        float total = 0.f;
        Vec model = Vec::Zero((Eigen::Index)depth);
        std::vector<float> spread(depth, 0.f);
        for (size_t j = 0; j < B; ++j) {
            const SomIndex bmu(*this, dataset.getLastBMU(j));
            const float w = static_cast<float>(calculateNeighbourhoodWeight(here.getX(), here.getY(), bmu.getX(), bmu.getY(), currentSigma));
            total += w;                              // Eq. 47
            const Vec delta = transform.Stepper(samples[j], model, valids[j]);
            requireLength(delta, depth, "Stepper");
            const Vec again = transform.Stepper(samples[j], model, valids[j]);    // :867 calls it on the model before the step
            const float c = w / total;
            for (size_t d = 0; d < depth; ++d) {
                const float step = c * delta[(Eigen::Index)d];
                const float sq = (w * again[(Eigen::Index)d]) * delta[(Eigen::Index)d];
                model[(Eigen::Index)d] = model[(Eigen::Index)d] + step;             // Eq. 53
                spread[d] = spread[d] + sq;                                         // Eq. 68
            }
        }
        for (size_t d = 0; d < depth; ++d) {
            newMap[node * depth + d] = model[(Eigen::Index)d];
            newSigma[node * depth + d] = std::sqrt(spread[d] / total);               // Eq. 69
        }
        hWeight[node] = total;
    }
    hMap.swap(newMap);
    hSigma.swap(newSigma);
    return mse;
}

// Som.cpp:716-754
void Som::hostTrainBatchSom(DataSet &data, size_t numberOfEpochs, double sigma0, double sigmaDecay, bool updateUMatrixAfterEpoch)
{
    {
        const std::lock_guard<std::mutex> lock(metricsMutex);   // (the reference resets without it: vsom_host.cpp, trainBatchSom)
        metrics = Som::Metrics(numberOfEpochs);
    }
    for (size_t i = 0; i < numberOfEpochs; ++i) {
        std::cout << "Training VSOM epoch " << i << "/" << numberOfEpochs << '\n';
        const double sigma = sigma0 * std::exp(-sigmaDecay * static_cast<double>(i));
        if (sigma < 1.0)
            return;
        float mse = 0.f;
        size_t chunks = 0;
        while (!data.hasReadWholeDataStream()) {
            data.loadNextDataFromStream();
            mse += hostBatchEpoch(data, sigma, i == 0);
            ++chunks;
        }
        mse /= static_cast<float>(chunks);
        {
            const std::lock_guard<std::mutex> lock(metricsMutex);
            metrics.MeanSquaredError[i] = mse;
        }
        data.resetStreamLoadPosition();
        if (updateUMatrixAfterEpoch)
            updateUMatrix(data.getWeights());        // :751-752 / :1183-1184
    }
}

// Som.cpp:885-947
Som::TrainingReturnValue Som::hostTrainSingle(const Vec &v, const Vec &valid, const Vec &weights, double eta, double sigma,
                                              size_t &lastBMU, WeigthDecayFunction fn)
{
    hostEnsure();
    const size_t bmu = sigma > SIGMA_SWITCH_TO_LOCAL ? hostFindBmu(v, valid, weights) : hostFindLocalBmu(v, valid, lastBMU, weights);
    const size_t bx = bmu % width, by = bmu / width;
    lastBMU = by * width + bx;
    // truncating window of +-2.5 sigma, clipped to [0, width] x [0, height] (:899-903)
    const size_t x0 = static_cast<size_t>(std::max(static_cast<double>(bx) - 2.5 * sigma, 0.));
    const size_t y0 = static_cast<size_t>(std::max(static_cast<double>(by) - 2.5 * sigma, 0.));
    const size_t x1 = static_cast<size_t>(std::min(static_cast<double>(bx) + 2.5 * sigma, static_cast<double>(width)));
    const size_t y1 = static_cast<size_t>(std::min(static_cast<double>(by) + 2.5 * sigma, static_cast<double>(height)));
    Vec validWeights((Eigen::Index)valid.size());
    for (Eigen::Index d = 0; d < valid.size(); ++d)
        validWeights[d] = valid[d] * weights[d];
    for (size_t y = y0; y < y1; ++y)
        for (size_t x = x0; x < x1; ++x) {
            const size_t n = y * width + x;
            float *M = &hMap[n * depth], *S = &hS[n * depth], *Sg = &hSigma[n * depth];
            const Vec delta = transform.Stepper(v, row(hMap, n, depth), validWeights);
            requireLength(delta, depth, "Stepper");
            const double h = calculateNeighbourhoodWeight(x, y, bx, by, sigma);
            if (fn == WeigthDecayFunction::Exponential) {
                hWeight[n] += static_cast<float>(h * eta);
                const float k = static_cast<float>(h * eta);
                for (size_t d = 0; d < depth; ++d)
                    M[d] = M[d] + k * delta[(Eigen::Index)d];
            } else {
                hWeight[n] += static_cast<float>(h);                                   // Eq. 47
                const double t = hWeight[n] == 0 ? 1.0 : h / hWeight[n];
                const Vec step = transform.Stepper(v, row(hMap, n, depth), validWeights);
                const float k = static_cast<float>(t);
                for (size_t d = 0; d < depth; ++d)
                    M[d] = M[d] + k * step[(Eigen::Index)d];                              // Eq. 53
            }
            const double norm = hWeight[n] == 0 ? 0.000001 : hWeight[n];
            const Vec after = transform.Stepper(v, row(hMap, n, depth), validWeights);  // on the UPDATED model (:941)
            const float hf = static_cast<float>(h), nf = static_cast<float>(norm);
            for (size_t d = 0; d < depth; ++d) {
                S[d] = S[d] + hf * (delta[(Eigen::Index)d] * after[(Eigen::Index)d]);    // Eq. 68
                Sg[d] = std::sqrt(std::fabs(S[d] / nf));                                  // Eq. 69
            }
        }
    Vec residual = transform.Comparer(v, row(hMap, bmu, depth), row(hSigma, bmu, depth), validWeights);
    const float dist = static_cast<float>(hostDist(bmu, v, valid, weights));
    return TrainingReturnValue{SomIndex(bx, by), residual, dist};
}

// Som.cpp:1135-1187
void Som::hostTrainBasicSom(DataSet &data, size_t numberOfEpochs, double eta0, double etaDecay, double sigma0, double sigmaDecay,
                            WeigthDecayFunction fn, bool updateUMatrixAfterEpoch)
{
    {
        const std::lock_guard<std::mutex> lock(metricsMutex);   // (the reference resets without it: vsom_host.cpp, trainBatchSom)
        metrics = Som::Metrics(numberOfEpochs);
    }
    const Vec weights = data.getWeights();
    for (size_t i = 0; i < numberOfEpochs; ++i) {
        const double eta = eta0 * std::exp(-etaDecay * static_cast<double>(i));
        double sigma = sigma0 * std::exp(-sigmaDecay * static_cast<double>(i));
        if (sigma < 1.0)
            sigma = 1.0;
        std::cout << "Epoch: " << i + 1 << "/" << numberOfEpochs << "\teta: " << eta << "\tsigma: " << sigma << "\n";
        float mse = 0.f;
        size_t chunks = 0;
        while (!data.hasReadWholeDataStream()) {
            data.loadNextDataFromStream();
            const size_t B = data.size();
            for (size_t j = 0; j < B; ++j) {
                auto r = hostTrainSingle(data.getData(j), asFloat(data.getValidity(j)), weights, eta, sigma, data.getLastBMU(j), fn);
                hHits[getIndex(r.bmu)] += 1;                                   // addBmu (:1189-1192)
                mse += packet_order_square_sum(r.residual) / static_cast<float>(B);
            }
            ++chunks;
        }
        mse /= static_cast<float>(chunks);
        {
            const std::lock_guard<std::mutex> lock(metricsMutex);
            metrics.MeanSquaredError[i] = mse;
        }
        data.resetStreamLoadPosition();
        if (updateUMatrixAfterEpoch)
            updateUMatrix(data.getWeights());        // :751-752 / :1183-1184
    }
    std::cout << "\rTraining SOM:100%\n";
}

// ---- consumers of the search outside the training loop, with the caller's hooks ----
// Som.cpp:313-332: node 0 seeds the search whatever its hit count; a later node replaces it only when it is
// strictly closer AND has at least minBmuHits hits
size_t Som::hostFindRestrictedBmu(const Vec &v, const Vec &valid, size_t minBmuHits, const Vec &weights) const
{
    hostEnsure();
    double minDist = hostDist(0, v, valid, weights);
    size_t minIndex = 0;
    for (size_t i = 0; i < width * height; ++i) {
        const double d = hostDist(i, v, valid, weights);
        if (d < minDist && hHits[i] >= minBmuHits) {
            minDist = d;
            minIndex = i;
        }
    }
    return minIndex;
}

// Som.cpp:143-157: sigma-normalised raw distance; no hook is involved (the reference computes M - v itself)
double Som::hostDistRaw(size_t pos, const Vec &v, const Vec &valid, const Vec &weights) const
{
    hostEnsure();
    Vec a((Eigen::Index)depth), b((Eigen::Index)depth);
    for (size_t d = 0; d < depth; ++d) {
        const float s = hSigma[pos * depth + d];
        const float sM = s < 0.00001f ? 0.00001f : s;                       // :150
        const float diff = hMap[pos * depth + d] - v[(Eigen::Index)d];
        const float validWeight = valid[(Eigen::Index)d] * weights[(Eigen::Index)d];   // :153
        a[(Eigen::Index)d] = diff / sM;
        const float t = diff * validWeight;
        b[(Eigen::Index)d] = t / sM;
    }
    return (double)packet_order_dot(a, b);                                   // :156
}
