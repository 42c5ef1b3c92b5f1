// host_api_test.cpp -- exercises the C++ mirror (vsom_api.hpp) the way the reference's perf harness
// drives libsom (tests/performance/perf_tests.cpp:74-140): construct, randomInitialize, train with
// each WeigthDecayFunction, findBmu / findLocalBmu / euclidianWeightedDist / trainSingle.
// Writes the resulting state to binary dumps that tests/test_gpu_host_cpp.py compares with the oracle.
//   usage: host_api_test <outdir>
#include "SOM.hpp"
#include "vsom_hip.h"
#include "DataSet.hpp"
#include "Transformation.hpp"
#include "MnistDataLoader.hpp"

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <memory>
#include <cstdlib>
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

// deterministic samples (same formula in the python test)
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

// `host_api_test perf`: the whole C++ path (loader -> DataSet -> pipelined Som::train) at BASELINE's size
static int perf()
{
    const size_t W = 128, H = 128, J = 784, NROWS = 16384, CHUNK = 4096;
    auto rows = make_rows(NROWS, J, 777u);
    ArrayDataLoader loader(rows.data(), NROWS, J, CHUNK);
    DataSet ds(loader);
    Som som{W, H, J};
    som.randomInitialize(42, 1);
    std::cout.setstate(std::ios_base::failbit);                      // the drivers print per epoch
    som.train(ds, 1, 0.0, 0.0, 40.0, 0.1, Som::WeigthDecayFunction::BatchMap);   // warm-up (first epoch: full search)
    const auto t0 = std::chrono::steady_clock::now();
    const size_t epochs = 4;
    som.train(ds, epochs, 0.0, 0.0, 40.0, 0.1, Som::WeigthDecayFunction::BatchMap);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout.clear();
    std::printf("{\"host_cpp_train\": \"128x128x784, %zu rows in chunks of %zu, %zu epochs (epoch 0 full search, later local)\", "
                "\"s_per_epoch\": %.4f, \"samples_per_s\": %.0f}\n", NROWS, CHUNK, epochs, dt / epochs, NROWS * epochs / dt);
    return 0;
}

// `host_api_test perf_tiny`: the reference's perf-harness scenario shape (perf_tests.cpp:74-112): 10x10 map,
// 20 nine-dimensional rows, 300 epochs, sigma0 10 / decay 0.01, eta0 0.001 / decay 0.01
static int perf_tiny()
{
    const size_t W = 10, H = 10, J = 9, NROWS = 20;
    auto rows = make_rows(NROWS, J, 99u);
    std::cout.setstate(std::ios_base::failbit);
    for (int mode = 0; mode < 2; ++mode) {
        ArrayDataLoader loader(rows.data(), NROWS, J);
        DataSet ds(loader);
        Som som{W, H, J};
        som.randomInitialize(1, 1);
        const auto fn = mode == 0 ? Som::WeigthDecayFunction::BatchMap : Som::WeigthDecayFunction::Exponential;
        som.train(ds, 5, 0.001, 0.01, 10.0, 0.01, fn);
        som.randomInitialize(1, 1);
        const auto t0 = std::chrono::steady_clock::now();
        som.train(ds, 300, 0.001, 0.01, 10.0, 0.01, fn);
        (void)som.getNeuron(size_t{0});
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        const int epochs = mode == 0 ? 231 : 300;     // batch training stops when sigma < 1 (Som.cpp:729-730)
        std::cout.clear();
        std::printf("{\"host_cpp_tiny\": \"%s 10x10x9, 20 rows\", \"us_per_epoch\": %.1f}\n", mode == 0 ? "BatchMap" : "Exponential",
                    dt / epochs * 1e6);
        std::cout.setstate(std::ios_base::failbit);
    }
    std::cout.clear();
    return 0;
}

// `host_api_test perf_mnist <folder>`: BASELINE configuration 3's data path end to end -- IDX files (60000 x 784,
// written by the caller) -> MnistDataLoader(4096) -> DataSet -> Som::train(BatchMap) on a 128x128 map
static int perf_mnist(const std::string &folder)
{
    MnistDataLoader loader(4096);
    loader.open(folder.c_str());
    DataSet ds(loader);
    Som som{128, 128, ds, Transformation::Standard(loader.getNames())};
    som.randomInitialize(42, 1);
    std::cout.setstate(std::ios_base::failbit);
    som.train(ds, 1, 0.0, 0.0, 40.0, 0.05, Som::WeigthDecayFunction::BatchMap);
    const auto t0 = std::chrono::steady_clock::now();
    const size_t epochs = 2;
    som.train(ds, epochs, 0.0, 0.0, 40.0, 0.05, Som::WeigthDecayFunction::BatchMap);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::cout.clear();
    std::printf("{\"host_cpp_mnist\": \"128x128x794, 60000 rows via MnistDataLoader(4096), %zu epochs\", \"s_per_epoch\": %.4f, "
                "\"samples_per_s\": %.0f}\n", epochs, dt / epochs, 60000.0 * epochs / dt);
    return 0;
}

// `host_api_test perf_e2e array <rows.f32> <nrows> <depth> <chunk>` / `perf_e2e mnist <folder> <chunk>`:
// Som::train(..., BatchMap) through the reference API on a 128x128 map, one JSON line per run with the wall time
// of the call, the device time of the same epochs (the library's per-phase HIP-event spans on the context's
// stream: staging, search, neighbourhood prefix, chains, expansion -- the side stream's bmuHits / MSE launch runs
// beside them) and their ratio.  Runs: three times ONE epoch (every epoch a full search: Som.cpp:738 passes
// i == 0) and a five-epoch schedule (one full-search epoch, four local).  VSOM_UPDATE_MODE selects the arithmetic.
static int perf_e2e(int argc, char **argv)
{
    const std::string kind = argv[2];
    std::unique_ptr<IDataLoader> loader;
    std::vector<float> rows;
    size_t nrows = 0, depth = 0, chunk = 4096;
    if (kind == "array" && argc > 6) {
        nrows = std::stoul(argv[4]);
        depth = std::stoul(argv[5]);
        chunk = std::stoul(argv[6]);
        rows.resize(nrows * depth);
        std::ifstream f(argv[3], std::ios::binary);
        f.read((char *)rows.data(), (std::streamsize)(rows.size() * 4));
        if (!f) {
            std::fprintf(stderr, "cannot read %s\n", argv[3]);
            return 2;
        }
        loader = std::make_unique<ArrayDataLoader>(rows.data(), nrows, depth, chunk);
    } else if (kind == "mnist" && argc > 4) {
        chunk = std::stoul(argv[4]);
        auto m = std::make_unique<MnistDataLoader>(chunk);
        m->open(argv[3]);
        nrows = 60000;
        loader = std::move(m);
    } else {
        std::fprintf(stderr, "usage: perf_e2e array <rows.f32> <nrows> <depth> <chunk> | perf_e2e mnist <folder> <chunk>\n");
        return 2;
    }
    DataSet ds(*loader);
    Som som{128, 128, ds, Transformation::Standard(loader->getNames())};
    som.randomInitialize(42, 1);
    const char *arith = std::getenv("VSOM_UPDATE_MODE");
    std::cout.setstate(std::ios_base::failbit);
    som.train(ds, 1, 0.0, 0.0, 40.0, 0.05, Som::WeigthDecayFunction::BatchMap);        // warm-up: allocations, code objects
    // host side alone: what one pass of the loader + DataSet staging costs (no device work in flight)
    double host_ms = 0;
    size_t host_chunks = 0;
    {
        ds.resetStreamLoadPosition();
        const auto h0 = std::chrono::steady_clock::now();
        do {
            ds.loadNextDataFromStream();
            ++host_chunks;
        } while (!ds.hasReadWholeDataStream());
        host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
        ds.resetStreamLoadPosition();
    }
    static const char *const names[VSOM_T_COUNT] = {"stage", "bmu", "finish", "cw", "update", "online", "sigma"};
    auto run = [&](const char *what, size_t epochs, double sigma0) {
        float ms[VSOM_T_COUNT];
        uint32_t cnt[VSOM_T_COUNT];
        vsom_get_timing(som.context(), ms, cnt, 1);
        vsom_enable_timing(som.context(), 1);
        const auto t0 = std::chrono::steady_clock::now();
        som.train(ds, epochs, 0.0, 0.0, sigma0, 0.05, Som::WeigthDecayFunction::BatchMap);
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        vsom_get_timing(som.context(), ms, cnt, 1);
        vsom_enable_timing(som.context(), 0);
        double dev = 0;
        for (int i = 0; i < VSOM_T_COUNT; ++i)
            if (i != VSOM_T_FINISH)
                dev += ms[i];
        std::cout.clear();
        std::printf("{\"e2e\": \"%s\", \"loader\": \"%s\", \"arithmetic\": \"%s\", \"rows\": %zu, \"depth\": %zu, \"chunk\": %zu, "
                    "\"epochs\": %zu, \"wall_ms_per_epoch\": %.3f, \"device_ms_per_epoch\": %.3f, \"device_over_wall\": %.4f, "
                    "\"samples_per_s\": %.0f, \"device_only_samples_per_s\": %.0f, \"host_load_ms_per_epoch\": %.3f, \"chunks_per_epoch\": %zu",
                    what, kind.c_str(), arith ? arith : "strict", nrows, som.getDepth(), chunk, epochs, wall / epochs, dev / epochs,
                    dev / wall, nrows * epochs / wall * 1e3, nrows * epochs / dev * 1e3, host_ms, host_chunks);
        for (int i = 0; i < VSOM_T_COUNT; ++i)
            if (cnt[i])
                std::printf(", \"%s_ms\": %.3f", names[i], ms[i] / epochs);
        std::printf("}\n");
        std::fflush(stdout);
        std::cout.setstate(std::ios_base::failbit);
    };
    for (int r = 0; r < 3; ++r)
        run("one epoch per call (full search)", 1, 40.0);
    run("five-epoch schedule (1 full + 4 local)", 5, 40.0);
    run("five-epoch schedule (1 full + 4 local)", 5, 40.0);
    std::cout.clear();
    return 0;
}

// `host_api_test ref_harness <fixture_rows.f32> [scale]`: every scenario of the reference's own performance harness
// (tests/performance/perf_tests.cpp:74-402) through the mirror's reference API, one JSON line each.  The harness's
// database fixture (the `ican` table: 20 rows x 9 columns) arrives as a binary row file (tests/golden/ican_fixture.json
// holds it as data); its per-call loops keep their counts except the two million-call ones, which run `scale` times
// fewer calls and report the time per call like the others.
static int ref_harness(const char *fixture, size_t scale)
{
    const size_t FR = 20, FJ = 9;
    std::vector<float> fx(FR * FJ);
    {
        std::ifstream f(fixture, std::ios::binary);
        f.read((char *)fx.data(), (std::streamsize)(fx.size() * 4));
        if (!f) {
            std::fprintf(stderr, "cannot read %s\n", fixture);
            return 2;
        }
    }
    using clk = std::chrono::steady_clock;
    auto us_since = [](clk::time_point t0) { return std::chrono::duration<double, std::micro>(clk::now() - t0).count(); };
    auto line = [](const char *scenario, const char *ref, size_t calls, double us_total, const char *unit) {
        std::cout.clear();
        std::printf("{\"scenario\": \"%s\", \"reference\": \"tests/performance/perf_tests.cpp:%s\", \"calls\": %zu, \"%s\": %.2f}\n",
                    scenario, ref, calls, unit, us_total / (double)calls);
        std::fflush(stdout);
        std::cout.setstate(std::ios_base::failbit);
    };
    std::cout.setstate(std::ios_base::failbit);          // the drivers print per epoch
    const size_t runs = 100;
    // ---- Som::randomInitialize (:162-178)
    {
        Som sut{100, 100, 100};
        sut.randomInitialize(1, 1);                      // (first call: allocations)
        const auto t0 = clk::now();
        sut.randomInitialize(1, 1);
        sut.updateUMatrix(Eigen::VectorXf::Random(100));
        auto a = sut.getUMatrix();
        (void)a;
        for (size_t i = 0; i < runs; ++i)
            sut.randomInitialize(1, 1);
        line("Som::randomInitialize() 100x100x100", "162-178", runs, us_since(t0), "us_per_call");
    }
    // ---- Som::train x 3 decay functions, 10x10 map on the 20 x 9 fixture, 300 epochs (:74-112, 390-392)
    const Som::WeigthDecayFunction fns[3] = {Som::WeigthDecayFunction::Exponential, Som::WeigthDecayFunction::BatchMap,
                                             Som::WeigthDecayFunction::InverseProportional};
    const char *fn_names[3] = {"Som::train(exponentialWeightDecay) 10x10x9, 20 rows, 300 epochs",
                               "Som::train(batchMap) 10x10x9, 20 rows, 300 epochs asked (sigma < 1 ends it after 231)",
                               "Som::train(inverseProportionalWeightDecay) 10x10x9, 20 rows, 300 epochs"};
    for (int m = 0; m < 3; ++m) {
        ArrayDataLoader loader(fx.data(), FR, FJ);
        DataSet ds(loader);
        Som sut{10, 10, ds, Transformation::Standard(loader.getNames())};
        sut.randomInitialize(7, 1);
        sut.train(ds, 3, 0.001, 0.01, 10.0, 0.01, fns[m]);          // warm-up: allocations, code objects
        sut.train(ds, 1, 0.001, 0.01, 1.0, 0.01, fns[m]);           // ... and those of the sigma <= 1 epochs (a kernel's first
                                                                    //     launch loads its code: ~15 ms, once per process)
        (void)sut.getNeuron(size_t{0});                             // ... and the state download's
        sut.randomInitialize(7, 1);
        const auto t0 = clk::now();
        sut.train(ds, 300, 0.001, 0.01, 10.0, 0.01, fns[m]);
        (void)sut.getNeuron(size_t{0});
        // the reference divides by the 300 epochs it asked for (:109-111)
        line(fn_names[m], "74-112", 300, us_since(t0), "us_per_epoch");
    }
    // ---- evaluate / measureSimilarity / variationalAutoEncoder on a 100x100 map over the fixture (:142-160, 297-318, 346-368)
    {
        ArrayDataLoader loader(fx.data(), FR, FJ);
        DataSet ds(loader);
        ds.loadNextDataFromStream();
        Som sut{100, 100, ds.vectorLength(), Transformation::Standard(ds.getNames())};
        sut.randomInitialize(7, 1);
        (void)sut.evaluate(ds);
        auto t0 = clk::now();
        for (size_t r = 0; r < runs; ++r)
            (void)sut.evaluate(ds);
        line("Som::evaluate() 100x100x9 over 20 rows", "142-160", runs, us_since(t0), "us_per_call");
        (void)sut.measureSimilarity(&ds, 1, 1);
        t0 = clk::now();
        for (size_t r = 0; r < runs; ++r)
            (void)sut.measureSimilarity(&ds, 1, 1);
        line("Som::measureSimilarity() 100x100x9 over 20 rows", "297-318", runs, us_since(t0), "us_per_call");
        (void)sut.variationalAutoEncoder(&ds, 0);
        t0 = clk::now();
        for (size_t r = 0; r < runs; ++r)
            (void)sut.variationalAutoEncoder(&ds, 0);
        line("Som::variationalAutoEncoder() 100x100x9 over 20 rows", "346-368", runs, us_since(t0), "us_per_call");
    }
    // ---- updateUMatrix on a 100x100 CombinatorialLinearRegression map of depth J(J-1) = 72 (:320-344)
    {
        ArrayDataLoader loader(fx.data(), FR, FJ);
        DataSet ds(loader);
        ds.loadNextDataFromStream();
        Som sut{100, 100, ds.vectorLength() * (ds.vectorLength() - 1), Transformation::CombinatorialLinearRegression(ds.getNames())};
        sut.randomInitialize(7, 1);
        sut.updateUMatrix(ds.getWeights());
        const auto t0 = clk::now();
        for (size_t r = 0; r < runs; ++r)
            sut.updateUMatrix(ds.getWeights());
        line("Som::updateUMatrix() 100x100 CLR map, depth 72", "320-344", runs, us_since(t0), "us_per_call");
    }
    // ---- the single-vector calls on a 100x100x100 map (:114-140, 181-295)
    {
        Som sut{100, 100, 100};
        sut.randomInitialize(7, 1);
        std::srand(12345);
        std::vector<Eigen::VectorXf> samples(1000);
        for (auto &v : samples)
            v = Eigen::VectorXf::Random(100);
        const Eigen::VectorXf ones = Eigen::VectorXf::Ones(100);
        const Eigen::VectorXf mv = Eigen::VectorXf::Random(100);
        std::vector<size_t> positions(1000);
        for (auto &e : positions)
            e = std::rand() % 100 * 100;                 // (the harness's expression: a multiple of 100 below 10000)
        for (int m = 0; m < 2; ++m) {
            const auto fn = m == 0 ? Som::WeigthDecayFunction::Exponential : Som::WeigthDecayFunction::InverseProportional;
            sut.randomInitialize(7, 1);
            size_t warm = 0;
            for (size_t i = 0; i < 10; ++i)
                (void)sut.trainSingle(samples[i], ones, ones, 0.1, 50, warm, fn);
            const auto t0 = clk::now();
            for (size_t i = 0; i < 1000; ++i)
                (void)sut.trainSingle(samples[i], ones, ones, 0.1, 50, positions[i], fn);
            line(m == 0 ? "Som::trainSingle(exponentialWeightDecay) 100x100x100, sigma 50"
                        : "Som::trainSingle(inverseProportionalWeightDecay) 100x100x100, sigma 50",
                 "114-140", 1000, us_since(t0), "us_per_call");
        }
        sut.randomInitialize(7, 1);
        {
            const size_t n = 1000000 / scale;
            double res = 0;
            (void)sut.euclidianWeightedDist(positions[0], mv, ones, ones);
            const auto t0 = clk::now();
            for (size_t i = 0; i < n; ++i)
                res += sut.euclidianWeightedDist(positions[i % 1000], mv, ones, ones) / 1000000;
            line("Som::euclidianWeightedDist() 100x100x100", "269-295", n, us_since(t0), "us_per_call");
            std::fprintf(stderr, "%g\n", res);
        }
        {
            (void)sut.findBmu(mv, ones, ones);
            const auto t0 = clk::now();
            for (size_t i = 0; i < 1000; ++i)
                (void)sut.findBmu(mv, ones, ones);
            line("Som::findBmu() 100x100x100", "181-199", 1000, us_since(t0), "us_per_call");
        }
        {
            const size_t n = 1000000 / scale;
            (void)sut.findLocalBmu(mv, ones, positions[0], ones);
            const auto t0 = clk::now();
            for (size_t i = 0; i < n; ++i)
                (void)sut.findLocalBmu(mv, ones, positions[i % 1000], ones);
            line("Som::findLocalBmu() 100x100x100", "200-222", n, us_since(t0), "us_per_call");
        }
        {
            (void)sut.findRestrictedBmu(mv, ones, 1, ones);
            auto t0 = clk::now();
            for (size_t i = 0; i < 1000; ++i)
                (void)sut.findRestrictedBmu(mv, ones, 1, ones);
            line("Som::findRestrictedBmu(minBmuHits 1) 100x100x100", "223-246", 1000, us_since(t0), "us_per_call");
            (void)sut.findRestrictedBmd(mv, ones, 0, ones);
            t0 = clk::now();
            for (size_t i = 0; i < 1000; ++i)
                (void)sut.findRestrictedBmd(mv, ones, 0, ones);
            line("Som::findRestrictedBmd(minBmuHits 0) 100x100x100", "247-268", 1000, us_since(t0), "us_per_call");
        }
    }
    std::cout.clear();
    return 0;
}

// `host_api_test threads <outdir>`: the boundary's threading contract (include/SOM.hpp:63,76: `_isTraining` is atomic,
// `metricsMutex` is public so that a second thread may read the metrics while train() blocks).
//   poll_batch.bin   Som::train(BatchMap) in a worker thread while this thread polls isTraining() and copies
//                    getMetrics() under metricsMutex
//   thr_a / thr_b    two Som objects on one device, trained at the same time from two host threads (one batch-map
//                    schedule, one online schedule)
// tests/test_gpu_robustness.py compares every dump with the oracle running the same schedules alone.
static int threads_mode(const std::string &out)
{
    const size_t W = 24, H = 20, J = 16, NROWS = 600, CHUNK = 200;
    auto rows_a = make_rows(NROWS, J, 4242u), rows_b = make_rows(NROWS, J, 777u);
    {
        ArrayDataLoader loader(rows_a.data(), NROWS, J, CHUNK);
        DataSet ds(loader);
        Som som{W, H, ds, Transformation::Standard(loader.getNames())};
        som.randomInitialize(42, 1);
        std::atomic<bool> done{false};
        std::thread worker([&] {
            som.train(ds, 6, 0.0, 0.0, 8.0, 0.2, Som::WeigthDecayFunction::BatchMap);
            done = true;
        });
        size_t polls = 0, seen_training = 0, sizes_ok = 1;
        while (!done) {
            if (som.isTraining())
                ++seen_training;
            {
                const std::lock_guard<std::mutex> lock(som.metricsMutex);
                const auto m = som.getMetrics();
                // the constructor sizes the metrics by the depth (Som.cpp:47), train() by the epochs (:719) -- nothing else
                if (m.MeanSquaredError.size() != J && m.MeanSquaredError.size() != 6)
                    sizes_ok = 0;
            }
            ++polls;
            std::this_thread::sleep_for(std::chrono::microseconds(50));
        }
        worker.join();
        std::cout << "poll: polls=" << polls << " seen_training=" << (seen_training ? 1 : 0) << " sizes_ok=" << sizes_ok
                  << " still_training=" << (som.isTraining() ? 1 : 0) << "\n";
        dump(out + "/poll_batch.bin", som, som.getMetrics().MeanSquaredError);
    }
    {
        ArrayDataLoader la(rows_a.data(), NROWS, J, CHUNK), lb(rows_b.data(), NROWS, J, CHUNK);
        DataSet da(la), db(lb);
        Som a{W, H, da, Transformation::Standard(la.getNames())};
        Som b{W, H, J, Transformation::StandardMedianEstimator({})};
        a.randomInitialize(5, 1);
        b.randomInitialize(6, 1);
        std::thread ta([&] { a.train(da, 5, 0.0, 0.0, 7.0, 0.25, Som::WeigthDecayFunction::BatchMap); });
        std::thread tb([&] { b.train(db, 3, 0.05, 0.1, 3.0, 0.3, Som::WeigthDecayFunction::Exponential); });
        ta.join();
        tb.join();
        dump(out + "/thr_a.bin", a, a.getMetrics().MeanSquaredError);
        dump(out + "/thr_b.bin", b, b.getMetrics().MeanSquaredError);
    }
    return 0;
}

// `host_api_test perf_e2e_online <rows.f32> <nrows> <depth> <chunk>`: Som::train(..., Exponential | InverseProportional)
// (Som.cpp:1135-1187: trainBasicSom, one trainSingle per sample) through the reference API on a 128x128 map at sigma 8,
// one JSON line per run: wall time of the call, device time of the same epochs (the library's HIP-event span around the
// chunk's kernels) and their ratio.
static int perf_e2e_online(int argc, char **argv)
{
    if (argc < 6) {
        std::fprintf(stderr, "usage: perf_e2e_online <rows.f32> <nrows> <depth> <chunk>\n");
        return 2;
    }
    const size_t nrows = std::stoul(argv[3]), depth = std::stoul(argv[4]), chunk = std::stoul(argv[5]);
    std::vector<float> rows(nrows * depth);
    {
        std::ifstream f(argv[2], std::ios::binary);
        f.read((char *)rows.data(), (std::streamsize)(rows.size() * 4));
        if (!f) {
            std::fprintf(stderr, "cannot read %s\n", argv[2]);
            return 2;
        }
    }
    ArrayDataLoader loader(rows.data(), nrows, depth, chunk);
    DataSet ds(loader);
    Som som{128, 128, ds, Transformation::Standard(loader.getNames())};
    som.randomInitialize(42, 1);
    std::cout.setstate(std::ios_base::failbit);
    static const char *const names[VSOM_T_COUNT] = {"stage", "bmu", "finish", "cw", "update", "online", "sigma"};
    auto run = [&](const char *what, Som::WeigthDecayFunction fn, size_t epochs) {
        float ms[VSOM_T_COUNT];
        uint32_t cnt[VSOM_T_COUNT];
        vsom_get_timing(som.context(), ms, cnt, 1);
        vsom_enable_timing(som.context(), 1);
        const auto t0 = std::chrono::steady_clock::now();
        som.train(ds, epochs, 0.1, 0.0, 8.0, 0.0, fn);               // eta 0.1, sigma 8 throughout
        const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        vsom_get_timing(som.context(), ms, cnt, 1);
        vsom_enable_timing(som.context(), 0);
        double dev = 0;
        for (int i = 0; i < VSOM_T_COUNT; ++i)
            dev += ms[i];
        uint64_t st[4] = {0, 0, 0, 0};
        vsom_get_online_search_stats(som.context(), st, 1);
        std::cout.clear();
        std::printf("{\"e2e\": \"%s\", \"loader\": \"array\", \"rows\": %zu, \"depth\": %zu, \"chunk\": %zu, \"epochs\": %zu, "
                    "\"wall_ms_per_epoch\": %.3f, \"device_ms_per_epoch\": %.3f, \"device_over_wall\": %.4f, \"samples_per_s\": %.0f, "
                    "\"device_only_samples_per_s\": %.0f, \"wall_us_per_sample\": %.3f, \"searched_through_the_image\": %s",
                    what, nrows, som.getDepth(), chunk, epochs, wall / epochs, dev / epochs, dev / wall, nrows * epochs / wall * 1e3,
                    nrows * epochs / dev * 1e3, wall / epochs / nrows * 1e3, st[0] ? "true" : "false");
        for (int i = 0; i < VSOM_T_COUNT; ++i)
            if (cnt[i])
                std::printf(", \"%s_ms\": %.3f", names[i], ms[i] / epochs);
        std::printf("}\n");
        std::fflush(stdout);
        std::cout.setstate(std::ios_base::failbit);
    };
    som.train(ds, 1, 0.1, 0.0, 8.0, 0.0, Som::WeigthDecayFunction::Exponential);      // warm-up: allocations, tables
    for (int r = 0; r < 2; ++r)
        run("Som::train(Exponential), sigma 8, one epoch per call", Som::WeigthDecayFunction::Exponential, 1);
    run("Som::train(Exponential), sigma 8, three epochs", Som::WeigthDecayFunction::Exponential, 3);
    run("Som::train(InverseProportional), sigma 8, one epoch per call", Som::WeigthDecayFunction::InverseProportional, 1);
    run("Som::train(InverseProportional), sigma 8, three epochs", Som::WeigthDecayFunction::InverseProportional, 3);
    std::cout.clear();
    return 0;
}

// `host_api_test mnist <folder> <outdir>`: BASELINE configuration 2's plumbing at test size -- IDX files
// -> MnistDataLoader (chunked) -> DataSet -> Som::train(BatchMap); the dump is compared with the oracle
// run on the same rows and chunk boundaries (tests/test_gpu_host_cpp.py)
static int mnist(const std::string &folder, const std::string &out)
{
    MnistDataLoader loader(256);
    loader.open(folder.c_str());
    DataSet ds(loader);
    Som som{12, 12, ds, Transformation::Standard(loader.getNames())};   // depth 794 from the loader
    som.randomInitialize(5, 1);
    som.train(ds, 3, 0.0, 0.0, 6.0, 0.2, Som::WeigthDecayFunction::BatchMap);
    dump(out + "/mnist_batch.bin", som, som.getMetrics().MeanSquaredError);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc > 1 && std::string(argv[1]) == "perf")
        return perf();
    if (argc > 1 && std::string(argv[1]) == "perf_tiny")
        return perf_tiny();
    if (argc > 2 && std::string(argv[1]) == "ref_harness")
        return ref_harness(argv[2], argc > 3 ? std::stoul(argv[3]) : 100);
    if (argc > 2 && std::string(argv[1]) == "threads")
        return threads_mode(argv[2]);
    if (argc > 2 && std::string(argv[1]) == "perf_mnist")
        return perf_mnist(argv[2]);
    if (argc > 2 && std::string(argv[1]) == "perf_e2e")
        return perf_e2e(argc, argv);
    if (argc > 2 && std::string(argv[1]) == "perf_e2e_online")
        return perf_e2e_online(argc, argv);
    if (argc > 3 && std::string(argv[1]) == "mnist")
        return mnist(argv[2], argv[3]);
    const std::string out = argc > 1 ? argv[1] : ".";
    const size_t W = 10, H = 10, J = 9, NROWS = 50, CHUNK = 20;
    auto rows = make_rows(NROWS, J, 12345u);

    // ---- batch map, standard transformation, two-and-a-half chunks per epoch ----
    {
        ArrayDataLoader loader(rows.data(), NROWS, J, CHUNK);
        DataSet ds(loader);
        Som som{W, H, ds, Transformation::Standard(loader.getNames())};
        som.randomInitialize(42, 1);
        std::cout << "group_members=" << (som.group() ? vsom_group_size(som.group()) : 1) << "\n";
        som.train(ds, 5, 0.001, 0.01, 10.0, 0.3, Som::WeigthDecayFunction::BatchMap);
        auto met = som.getMetrics();
        dump(out + "/batch_std.bin", som, met.MeanSquaredError);
        std::cout << "batch_std hits0=" << som.getBmuHits()[0] << " neuron0[0]=" << som.getNeuron(size_t{0})[0] << "\n";
    }
    // ---- batch map on a DataSet whose single chunk the caller loaded BEFORE train(): the first epoch finds
    //      the stream already read to its end, processes nothing and records 0/0 (Som.cpp:735-749); the
    //      following epochs reload and train (local search: only epoch 0 passes isFirst) ----
    {
        ArrayDataLoader loader(rows.data(), NROWS, J);
        DataSet ds(loader);
        Som som{W, H, ds, Transformation::Standard(loader.getNames())};
        som.randomInitialize(42, 1);
        ds.loadNextDataFromStream();
        som.train(ds, 3, 0.0, 0.0, 6.0, 0.2, Som::WeigthDecayFunction::BatchMap);
        dump(out + "/batch_preloaded.bin", som, som.getMetrics().MeanSquaredError);
    }
    // ---- online epochs, then batch-map epochs on the SAME map (with several devices the online path trains
    //      member 0 only and the other members take its state over before the sharded epochs) ----
    {
        ArrayDataLoader loader(rows.data(), NROWS, J, CHUNK);
        DataSet ds(loader);
        Som som{W, H, ds, Transformation::Standard(loader.getNames())};
        som.randomInitialize(13, 1);
        som.train(ds, 2, 0.05, 0.1, 4.0, 0.3, Som::WeigthDecayFunction::Exponential);
        auto mse = som.getMetrics().MeanSquaredError;
        som.train(ds, 2, 0.0, 0.0, 5.0, 0.2, Som::WeigthDecayFunction::BatchMap);
        for (float v : som.getMetrics().MeanSquaredError)
            mse.push_back(v);
        dump(out + "/online_then_batch.bin", som, mse);
    }
    // ---- online, exponential decay, median estimator ----
    {
        ArrayDataLoader loader(rows.data(), NROWS, J, CHUNK);
        DataSet ds(loader);
        Som som{W, H, J, Transformation::StandardMedianEstimator({})};
        som.randomInitialize(7, 1);
        som.train(ds, 3, 0.05, 0.1, 3.0, 0.5, Som::WeigthDecayFunction::Exponential);
        dump(out + "/online_median.bin", som, som.getMetrics().MeanSquaredError);
    }
    // ---- online schedules that run on into sigma <= 1 (Som.cpp:1148-1149 clamps sigma at 1, :891 then walks from lastBMU),
    //      the shape of the reference's own training scenario (perf_tests.cpp:74-112): two and a half chunks per epoch ----
    for (int m = 0; m < 2; ++m) {
        ArrayDataLoader loader(rows.data(), NROWS, J, CHUNK);
        DataSet ds(loader);
        Som som{W, H, J, m == 0 ? Transformation::Standard({}) : Transformation::StandardMedianEstimator({})};
        som.randomInitialize(11, 1);
        som.train(ds, 12, 0.05, 0.1, 3.0, 0.25, m == 0 ? Som::WeigthDecayFunction::InverseProportional : Som::WeigthDecayFunction::Exponential);
        dump(out + (m == 0 ? "/online_to_local_std.bin" : "/online_to_local_median.bin"), som, som.getMetrics().MeanSquaredError);
    }
    // ---- online, inverse proportional, CLR through the depth constructor (perf_tests.cpp:338-339) ----
    {
        const size_t Jc = 5;
        auto crows = make_rows(30, Jc, 777u);
        ArrayDataLoader loader(crows.data(), 30, Jc);
        DataSet ds(loader);
        auto t = Transformation::CombinatorialLinearRegression({});
        Som som{6, 6, t.Length(Jc), t};
        som.randomInitialize(3, 1);
        som.train(ds, 2, 0.01, 0.0, 2.0, 0.2, Som::WeigthDecayFunction::InverseProportional);
        dump(out + "/online_clr.bin", som, som.getMetrics().MeanSquaredError);
    }
    // ---- searches and trainSingle ----
    {
        Som som{W, H, J};
        som.randomInitialize(11, 1);
        Eigen::VectorXf v(J), ones = Eigen::VectorXf::Ones(J);
        for (size_t d = 0; d < J; ++d)
            v[d] = rows[d];
        SomIndex b = som.findBmu(v);
        SomIndex l = som.findLocalBmu(v, ones, 37, ones);
        double dist = som.euclidianWeightedDist(b, v, ones, ones);
        size_t last = 5;
        auto [pos, residual, derr] = som.trainSingle(v, ones, ones, 0.1, 2.0, last, Som::WeigthDecayFunction::Exponential);
        std::ofstream f(out + "/search.txt");
        f << som.getIndex(b) << " " << som.getIndex(l) << " " << std::hexfloat << dist << " " << som.getIndex(pos) << " "
          << last << " " << std::hexfloat << (double)derr << " " << (double)residual[0] << "\n";
        dump(out + "/single.bin", som, {});
        // a copy carries the state (copy constructor, SOM.hpp:90-105)
        Som copy{som};
        dump(out + "/single_copy.bin", copy, {});
        som.saveBinary((out + "/ckpt.vsom").c_str());             // lossless (incl. SMap)
        Som loaded{(out + "/ckpt.vsom").c_str()};
        dump(out + "/single_loaded.bin", loaded, {});
        som.addBmu(SomIndex(3, 4));
        som.addBmu(SomIndex(3, 4));
        som.updateUMatrix(ones);
        dump(out + "/single_text_src.bin", som, {});
        som.save((out + "/ckpt.txt").c_str());                    // the reference's Octave text format
        Som fromText{(out + "/ckpt.txt").c_str()};                // Som(const char*) sizes the map from the file
        dump(out + "/single_text_loaded.bin", fromText, {});
        fromText.save((out + "/ckpt2.txt").c_str());              // load -> save is idempotent
        {
            std::ofstream u(out + "/umatrix.txt");
            const UMatrix um = som.getUMatrix();   // keep the temporary alive for the loop
            for (double x : um.getData())
                u << std::hexfloat << x << "\n";
        }
    }
    // ---- consumers of the search: restricted BMU / BMD, U-matrix, evaluate, measureSimilarity ----
    {
        ArrayDataLoader loader(rows.data(), NROWS, J);
        DataSet ds(loader);
        Som som{W, H, ds, Transformation::Standard(loader.getNames())};
        som.randomInitialize(21, 1);
        som.train(ds, 2, 0.0, 0.0, 4.0, 0.2, Som::WeigthDecayFunction::BatchMap);   // gives hits, sigma
        ds.loadNextDataFromStream();
        Eigen::VectorXf v(J), ones = Eigen::VectorXf::Ones(J);
        for (size_t d = 0; d < J; ++d)
            v[d] = rows[3 * J + d];
        std::ofstream f(out + "/next_rows.txt");
        f << som.getIndex(som.findRestrictedBmu(v, ones, 1, ones)) << " " << som.getIndex(som.findRestrictedBmu(v, ones, 3, ones))
          << " " << som.getIndex(som.findRestrictedBmu(v, ones, 1000, ones)) << "\n";
        auto bmd = som.findRestrictedBmd(v, ones, 1, ones);
        std::ofstream fb(out + "/bmd.bin", std::ios::binary);
        fb.write((const char *)bmd.data(), bmd.size() * 8);
        som.updateUMatrix(ones);
        auto um = som.getUMatrix().getData();
        std::ofstream fu(out + "/umatrix.bin", std::ios::binary);
        fu.write((const char *)um.data(), um.size() * 8);
        f << std::hexfloat << som.euclidianWeightedDistRaw(17, v, ones, ones) << " " << som.evaluate(ds) << " "
          << som.measureSimilarity(&ds, 3, 1) << " " << som.measureSimilarity(&ds, 1000000, 1) << "\n";
        size_t drawn = som.variationalAutoEncoder(&ds, 1);
        f << (drawn < W * H ? 1 : 0) << "\n";
        dump(out + "/next_state.bin", som, {});
        // CLR U-matrix (perf_tests.cpp:335-352 runs updateUMatrix on a CLR map)
        auto t = Transformation::CombinatorialLinearRegression({});
        Som clr{5, 4, t.Length(4), t};
        clr.randomInitialize(5, 1);
        std::vector<float> sg(20 * 12);
        for (size_t k = 0; k < sg.size(); ++k)
            sg[k] = 0.25f + 0.01f * (float)(k % 7);
        clr.setState(nullptr, sg.data(), nullptr, nullptr, nullptr);
        clr.updateUMatrix(Eigen::VectorXf::Ones(12));
        auto umc = clr.getUMatrix().getData();
        std::ofstream fc(out + "/umatrix_clr.bin", std::ios::binary);
        fc.write((const char *)umc.data(), umc.size() * 8);
    }
    // ---- updateUMatrixAfterEpoch (Som.cpp:751-752): the U-matrix of the LAST epoch's map.  Under a group the
    //      sigmaMap rows of the other shards arrive on a second stream: updateUMatrix must join them first ----
    {
        ArrayDataLoader loader(rows.data(), NROWS, J, CHUNK);
        DataSet ds(loader);
        Som som{W, H, ds, Transformation::Standard(loader.getNames())};
        som.randomInitialize(33, 1);
        som.train(ds, 3, 0.0, 0.0, 5.0, 0.2, Som::WeigthDecayFunction::BatchMap, true);
        auto um = som.getUMatrix().getData();
        std::ofstream fu(out + "/umatrix_after_epoch.bin", std::ios::binary);
        fu.write((const char *)um.data(), um.size() * 8);
        dump(out + "/umatrix_after_epoch_state.bin", som, {});
    }
    // ---- a custom std::function transformation cannot run on the device: that Som lives on the host
    //      (src/vsom_custom.cpp; CPU test tests/test_host_custom.py), consumers outside training included ----
    {
        Transformation custom{.Comparer = [](const Eigen::VectorXf &, const Eigen::VectorXf &m, const Eigen::VectorXf &,
                                             const Eigen::VectorXf &) { return m; }};
        ArrayDataLoader loader(rows.data(), NROWS, J, CHUNK);
        DataSet ds(loader);
        Som som{W, H, ds, custom};
        som.randomInitialize(4, 1);
        som.train(ds, 1, 0.1, 0.1, 3.0, 0.1, Som::WeigthDecayFunction::BatchMap);
        const bool trained = som.getWeigthMap()[0] > 0.f && som.context() == nullptr;
        bool ran = true;
        try {
            som.updateUMatrix(Eigen::VectorXf::Ones(J));     // Som.cpp:999-1111 through the host path
            ds.loadNextDataFromStream();
            (void)som.evaluate(ds);
        } catch (const std::exception &e) {
            ran = false;
        }
        std::cout << "custom_transformation_host_path=" << (trained ? 1 : 0) << " consumers_run=" << (ran ? 1 : 0)
                  << " kind=" << custom.kind() << "\n";
    }
    std::cout << "host_api_test done\n";
    return 0;
}
