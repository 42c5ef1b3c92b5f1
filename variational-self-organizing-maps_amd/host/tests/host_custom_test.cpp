// host_custom_test -- the host path for caller-supplied Transformation hooks (src/vsom_custom.cpp).  Needs
// no GPU: a Som built with custom std::functions never creates a device context.
//   host_custom_test <outdir>
// 1. the reference's "Fakes" transformation (tests/test1.cpp:46-90: designated initialisers, Comparer =
//    Stepper = A.*X + B with P = [A | B]) called the way that test calls it; values go to fakes.txt;
// 2. caller-written lambdas that compute what Standard / StandardMedianEstimator compute, trained through
//    Som::train (batch map: two chunks, first + local-search epochs; online: Exponential and
//    InverseProportional) -- tests/test_host_custom.py compares every dump with the oracle bit for bit.
#include "SOM.hpp"
#include "DataSet.hpp"
#include "Transformation.hpp"

#include <fstream>
#include <iostream>
#include <string>
#include <vector>

static void dump(const std::string &path, const Som &som, const std::vector<float> &mse)
{
    const size_t N = som.getWidth() * som.getHeight(), D = som.getDepth();
    std::vector<float> m(N * D), s(N * D), S(N * D), w(N);
    std::vector<uint64_t> h(N);
    som.getState(m.data(), s.data(), S.data(), w.data(), h.data());
    std::ofstream f(path, std::ios::binary);
    uint64_t hdr[3] = {N, D, mse.size()};
    f.write((const char *)hdr, sizeof(hdr));
    f.write((const char *)m.data(), m.size() * 4);
    f.write((const char *)s.data(), s.size() * 4);
    f.write((const char *)S.data(), S.size() * 4);
    f.write((const char *)w.data(), w.size() * 4);
    f.write((const char *)h.data(), h.size() * 8);
    f.write((const char *)mse.data(), mse.size() * 4);
}

static std::vector<float> make_rows(size_t n, size_t d, unsigned seed)
{
    std::vector<float> r(n * d);
    unsigned s = seed;
    for (auto &v : r) {
        s = s * 1664525u + 1013904223u;
        v = (float)((s >> 8) & 0xFFFF) / 65536.0f * 2.0f - 1.0f;
    }
    return r;
}

using V = Eigen::VectorXf;

static Transformation likeStandard()
{
    return Transformation{.Comparer = [](const V &value, const V &model, const V &, const V &) { return model - value; },
                          .Stepper = [](const V &value, const V &model, const V &) { return value - model; }};
}

static Transformation likeMedian()
{
    return Transformation{.Comparer = [](const V &value, const V &model, const V &, const V &) { return model - value; },
                          .Stepper = [](const V &value, const V &model, const V &) {
                              V d = value - model;
                              for (Eigen::Index i = 0; i < d.size(); ++i) {
                                  const float a = d[i];
                                  d[i] = (a != a) ? a : (float)((a > 0.f) - (a < 0.f));
                              }
                              return d;
                          }};
}

int main(int argc, char **argv)
{
    const std::string out = argc > 1 ? argv[1] : ".";
    {   // ---- 1. tests/test1.cpp "Fakes" ----
        V P(8), X(4);
        P << 1, 2, 3, 4, 5, 6, 7, 8;
        X << 1, 2, 3, 4;
        auto names = std::vector<std::string>{};
        auto affine = [](const V &X, const V &P) {
            V Y(X.size());
            for (Eigen::Index i = 0; i < X.size(); ++i)
                Y[i] = P[i] * X[i] + P[P.size() / 2 + i];
            return Y;
        };
        auto transformation = Transformation{.Comparer = [affine](const V &X, const V &P, const V &, const V &) { return affine(X, P); },
                                             .Stepper = [affine](const V &X, const V &P, const V &) { return affine(X, P); },
                                             .Displayer = [&names](const V &) { return names; }};
        std::ofstream f(out + "/fakes.txt");
        f << transformation.kind() << "\n" << transformation.Comparer(X, P, P, X) << "\n" << transformation.Stepper(X, P, X) << "\n"
          << transformation.Length(4) << " " << transformation.Name << "\n";
        // a Som built with it keeps its state on the host: constructible, accessors work, no device needed
        Som som{3, 2, 8, transformation};
        som.randomInitialize(5, 1);
        f << som.getNeuron(size_t{0}).size() << " " << som.getWidth() << " " << som.getHeight() << " " << (som.context() == nullptr) << "\n";
    }
    const size_t W = 10, H = 10, J = 9, NROWS = 50, CHUNK = 20;
    auto rows = make_rows(NROWS, J, 12345u);
    std::cout.setstate(std::ios_base::failbit);
    {   // ---- 2a. batch map with Standard-equivalent hooks: 3 chunks per epoch, first + local epochs ----
        ArrayDataLoader loader(rows.data(), NROWS, J, CHUNK);
        DataSet ds(loader);
        Som som{W, H, ds, likeStandard()};
        som.randomInitialize(42, 1);
        som.train(ds, 5, 0.001, 0.01, 10.0, 0.3, Som::WeigthDecayFunction::BatchMap);
        dump(out + "/custom_batch_std.bin", som, som.getMetrics().MeanSquaredError);
        // searches and the distance on the trained map
        V v(J), ones = V::Ones(J);
        for (size_t d = 0; d < J; ++d)
            v[(Eigen::Index)d] = rows[7 * J + d];
        std::ofstream f(out + "/custom_search.txt");
        f << som.getIndex(som.findBmu(v)) << " " << som.getIndex(som.findLocalBmu(v, ones, 37, ones)) << " " << std::hexfloat
          << som.euclidianWeightedDist(som.findBmu(v), v, ones, ones) << "\n";
        // ---- 2a'. the consumers outside training, through the same hooks (Som.cpp:143-157, 313-332, 457-523,
        //           631-714, 999-1111): restricted BMU / BMD, raw distance, U-matrix, evaluate, measureSimilarity ----
        std::ofstream g(out + "/custom_consumers.txt");
        g << som.getIndex(som.findRestrictedBmu(v, ones, 1, ones)) << " " << som.getIndex(som.findRestrictedBmu(v, ones, 2, ones))
          << " " << som.getIndex(som.findRestrictedBmu(v, ones, 1000, ones)) << "\n";
        auto bmd = som.findRestrictedBmd(v, ones, 1, ones);
        std::ofstream fb(out + "/custom_bmd.bin", std::ios::binary);
        fb.write((const char *)bmd.data(), bmd.size() * 8);
        som.updateUMatrix(ones);
        auto um = som.getUMatrix().getData();
        std::ofstream fu(out + "/custom_umatrix.bin", std::ios::binary);
        fu.write((const char *)um.data(), um.size() * 8);
        ArrayDataLoader whole(rows.data(), NROWS, J);
        DataSet all(whole);
        all.loadNextDataFromStream();
        g << std::hexfloat << som.euclidianWeightedDistRaw(17, v, ones, ones) << " " << som.evaluate(all) << " "
          << som.measureSimilarity(&all, 3, 1) << " " << som.measureSimilarity(&all, 1000000, 1) << "\n";
        const size_t drawn = som.variationalAutoEncoder(&all, 1);
        g << (drawn < W * H ? 1 : 0) << "\n";
        // a copy of a custom-hook Som carries the trained state (the reference's implicit copy does)
        Som copy{som};
        Som assigned{2, 2, 3, likeStandard()};
        assigned = som;
        dump(out + "/custom_copy.bin", copy, som.getMetrics().MeanSquaredError);
        dump(out + "/custom_assigned.bin", assigned, som.getMetrics().MeanSquaredError);
        // Octave text checkpoint of a custom-hook Som: save -> load into a fresh one
        som.save((out + "/custom_ckpt.txt").c_str());
        Som loaded{W, H, J, likeStandard()};
        loaded.load((out + "/custom_ckpt.txt").c_str());
        g << std::defaultfloat << loaded.getNeuron(size_t{5})[2] << " " << loaded.getWeigthMap()[5] << "\n";
    }
    {   // ---- 2b. batch map with Median-equivalent hooks ----
        ArrayDataLoader loader(rows.data(), NROWS, J, CHUNK);
        DataSet ds(loader);
        Som som{W, H, ds, likeMedian()};
        som.randomInitialize(9, 1);
        som.train(ds, 3, 0.0, 0.0, 6.0, 0.2, Som::WeigthDecayFunction::BatchMap);
        dump(out + "/custom_batch_median.bin", som, som.getMetrics().MeanSquaredError);
    }
    {   // ---- 2c. online, Exponential, Median-equivalent hooks (as host_api_test's online_median case) ----
        ArrayDataLoader loader(rows.data(), NROWS, J, CHUNK);
        DataSet ds(loader);
        Som som{W, H, J, likeMedian()};
        som.randomInitialize(7, 1);
        som.train(ds, 3, 0.05, 0.1, 3.0, 0.5, Som::WeigthDecayFunction::Exponential);
        dump(out + "/custom_online_median.bin", som, som.getMetrics().MeanSquaredError);
    }
    {   // ---- 2d. online, InverseProportional, Standard-equivalent hooks, sigma reaching 1 (local search) ----
        ArrayDataLoader loader(rows.data(), NROWS, J);
        DataSet ds(loader);
        Som som{W, H, J, likeStandard()};
        som.randomInitialize(3, 1);
        som.train(ds, 3, 0.01, 0.0, 2.0, 0.7, Som::WeigthDecayFunction::InverseProportional);
        dump(out + "/custom_online_inv.bin", som, som.getMetrics().MeanSquaredError);
    }
    std::cout.clear();
    std::cout << "host_custom_test done\n";
    return 0;
}
