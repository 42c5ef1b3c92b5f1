// eigen_crosscheck.cpp -- TEST INFRASTRUCTURE (oracle/): replays one golden case on the REAL reference.
//
// What it is for: the oracle's two recalled conventions --
// Eigen's fp32 reduction order in `dot`/`squaredNorm` and `sign(NaN)` -- and every other arithmetic detail
// are pinned the day this program runs.  It links the reference's OWN translation units
// (/root/reference/src/{Som,Transformation,SomIndex,UMatrix,DataSet}.cpp, compiled where they lie by
// oracle/Makefile target `eigen_crosscheck`, outputs only under oracle/_ref/) against REAL Eigen 3
// (<Eigen/Dense> on the include path; no stand-in headers) and dumps lastBMU / map / sigmaMap / SMap /
// weightMap / bmuHits / MSE for the inputs of a committed golden (tests/golden/*.npz, handed over as a
// flat binary by tests/test_eigen_crosscheck.py, which compares the dump with the golden bit for bit).
// The build image has no Eigen, so today the recipe reports "Eigen absent" and the test skips with that
// reason; nothing here is shipped, imported by the product, or sent to the GPU box in source form
// (reference sources never leave /root/reference).
//
// Case file (little endian): int64 W,H,J,transform(0 std,1 median,2 clr),mode(0 batch,1 online),epochs,
// decayFn(0 exp,1 inv),nchunks ; double sigma0,sigmaDecay,eta,sigma ; int64 chunk_off[nchunks+1] ;
// float init_map[N*D] ; float X[B*J].        D = Transformation::Length(J), B = chunk_off[nchunks].
// Dump: batch -> per epoch {uint64 lastbmu[B], float map[N*D], sigma[N*D], weight[N], float mse} then
// uint64 hits[N]; online -> float map, sigma, S [N*D], weight[N], uint64 hits[N], lastbmu[B], float mse.
#if __has_include(<Eigen/Dense>)

#include "SOM.hpp"   // the reference's header (-I /root/reference/include)

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

namespace {

// in-memory IDataLoader (the reference ships only SQLite / MNIST loaders): rows of `depth` floats, chunks
// given by explicit offsets, wraps to the start after the last chunk like SqliteDataLoader.cpp:465-479
class MemLoader : public IDataLoader {
public:
    MemLoader(const std::vector<float> &x, size_t depth, std::vector<int64_t> off)
        : m_x(x), m_depth(depth), m_off(std::move(off)), m_weights(depth, 1.0f), m_binary(depth, 0), m_cont(depth, 1)
    {
        for (size_t d = 0; d < depth; ++d)
            m_names.push_back("c" + std::to_string(d));
    }
    size_t load() override
    {
        data.clear();
        const size_t c = m_currentIndex;
        for (int64_t r = m_off[c]; r < m_off[c + 1]; ++r) {
            RowData row;
            row.values = Eigen::Map<const Eigen::VectorXf>(m_x.data() + (size_t)r * m_depth, (Eigen::Index)m_depth);
            row.valid.assign(m_depth, 1);
            data.push_back(row);
        }
        m_currentIndex = (c + 2 < m_off.size()) ? c + 1 : 0;
        return data.size();
    }
    std::vector<RowData> getPreview(size_t) override { return {}; }
    bool open(const char *) override { return true; }
    std::vector<std::string> findAllColumns() override { return m_names; }
    void setColumnSpec(const std::vector<ColumnSpec>) noexcept override {}
    const std::vector<ColumnSpec> getColumnSpec() noexcept override { return {}; }
    float getWeight(size_t i) override { return m_weights[i]; }
    const std::vector<float> getWeights() const noexcept override { return m_weights; }
    const std::vector<int> &getBinary() const noexcept override { return m_binary; }
    const std::vector<int> &getContinuous() const noexcept override { return m_cont; }
    std::string getName(size_t i) const noexcept override { return m_names[i]; }
    const std::vector<std::string> getNames() const noexcept override { return m_names; }
    size_t getDepth() const noexcept override { return m_depth; }
    bool isAtStartOfDataStream() const noexcept override { return m_currentIndex == 0; }

private:
    const std::vector<float> &m_x;
    size_t m_depth;
    std::vector<int64_t> m_off;
    std::vector<float> m_weights;
    std::vector<int> m_binary, m_cont;
    std::vector<std::string> m_names;
};

// the reference keeps its state protected (SOM.hpp:41-63): a derived class reads and seeds it losslessly
struct Probe : Som {
    using Som::Som;
    void seed(const std::vector<float> &init, size_t D)
    {
        for (size_t i = 0; i < map.size(); ++i)
            map[i] = Eigen::Map<const Eigen::VectorXf>(init.data() + i * D, (Eigen::Index)D);
    }
    void dumpRows(std::ofstream &f, const std::vector<Eigen::VectorXf> &rows) const
    {
        for (const auto &r : rows)
            f.write((const char *)r.data(), (std::streamsize)(r.size() * sizeof(float)));
    }
    void dumpBatch(std::ofstream &f) const
    {
        dumpRows(f, map);
        dumpRows(f, sigmaMap);
        f.write((const char *)weightMap.data(), (std::streamsize)(weightMap.size() * sizeof(float)));
    }
    void dumpHits(std::ofstream &f) const
    {
        for (size_t h : bmuHits) {
            const uint64_t v = h;
            f.write((const char *)&v, 8);
        }
    }
    void dumpOnline(std::ofstream &f) const
    {
        dumpRows(f, map);
        dumpRows(f, sigmaMap);
        dumpRows(f, SMap);
        f.write((const char *)weightMap.data(), (std::streamsize)(weightMap.size() * sizeof(float)));
        dumpHits(f);
    }
    // one chunk of trainBasicSom's sample loop (Som.cpp:1161-1171) with the accumulator of :1153,1167
    float onlineChunk(DataSet &data, double eta, double sigma, WeigthDecayFunction fn, std::vector<uint64_t> &lastOut)
    {
        float mse = 0.0f;
        const size_t B = data.size();
        const Eigen::VectorXf weights = data.getWeights();
        for (size_t j = 0; j < B; ++j) {
            auto r = trainSingle(data.getData(j), data.getValidity(j).cast<float>(), weights, eta, sigma, data.getLastBMU(j), fn);
            addBmu(r.bmu);
            mse += r.residual.squaredNorm() / static_cast<float>(B);
        }
        lastOut.assign(B, 0);
        for (size_t j = 0; j < B; ++j)
            lastOut[j] = data.getLastBMU(j);
        return mse;
    }
};

template <typename T>
std::vector<T> readN(std::ifstream &f, size_t n)
{
    std::vector<T> v(n);
    f.read((char *)v.data(), (std::streamsize)(n * sizeof(T)));
    if (!f) {
        std::fprintf(stderr, "eigen_crosscheck: case file truncated\n");
        std::exit(3);
    }
    return v;
}

}   // namespace

int main(int argc, char **argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: eigen_crosscheck case.bin dump.bin\n");
        return 2;
    }
    std::ifstream in(argv[1], std::ios::binary);
    if (!in) {
        std::fprintf(stderr, "eigen_crosscheck: cannot open %s\n", argv[1]);
        return 2;
    }
    const auto hdr = readN<int64_t>(in, 8);
    const auto par = readN<double>(in, 4);
    const size_t W = (size_t)hdr[0], H = (size_t)hdr[1], J = (size_t)hdr[2];
    const int tr = (int)hdr[3], mode = (int)hdr[4];
    const size_t epochs = (size_t)hdr[5], nchunks = (size_t)hdr[7];
    const auto fn = hdr[6] == 0 ? Som::WeigthDecayFunction::Exponential : Som::WeigthDecayFunction::InverseProportional;
    const auto off = readN<int64_t>(in, nchunks + 1);
    const Transformation T = tr == 1   ? Transformation::StandardMedianEstimator({})
                             : tr == 2 ? Transformation::CombinatorialLinearRegression({})
                                       : Transformation::Standard({});
    const size_t D = T.Length(J), N = W * H, B = (size_t)off[nchunks];
    const auto init = readN<float>(in, N * D);
    const auto X = readN<float>(in, B * J);

    Probe som{W, H, D, T};      // the depth constructor, as perf_tests.cpp:338-339 builds a CLR map
    som.seed(init, D);
    MemLoader loader(X, J, off);
    DataSet data(loader);
    std::ofstream out(argv[2], std::ios::binary);
    std::cout.setstate(std::ios::failbit);      // the reference prints progress to stdout

    if (mode == 0) {
        // make_goldens.py batch_case: per epoch sigma = sigma0*exp(-decay*ep), stop below 1 (Som.cpp:727-730);
        // per chunk loadNextDataFromStream + trainBatchSomEpoch (Som.cpp:735-741); MSE = mean over chunks (:743)
        for (size_t ep = 0; ep < epochs; ++ep) {
            const double sigma = par[0] * std::exp(-par[1] * static_cast<double>(ep));
            if (sigma < 1.0)
                break;
            float mse = 0.0f;
            std::vector<uint64_t> lastAll;
            size_t chunks = 0;
            while (!data.hasReadWholeDataStream()) {
                data.loadNextDataFromStream();
                mse += som.trainBatchSomEpoch(data, sigma, ep == 0);
                for (size_t j = 0; j < data.size(); ++j)
                    lastAll.push_back((uint64_t)data.getLastBMU(j));
                ++chunks;
            }
            mse /= static_cast<float>(chunks);
            data.resetStreamLoadPosition();
            out.write((const char *)lastAll.data(), (std::streamsize)(lastAll.size() * 8));
            som.dumpBatch(out);
            out.write((const char *)&mse, 4);
        }
        som.dumpHits(out);
    } else {
        // make_goldens.py online_case: one chunk, B sequential trainSingle + addBmu (Som.cpp:1161-1171)
        data.loadNextDataFromStream();
        std::vector<uint64_t> last;
        const float mse = som.onlineChunk(data, par[2], par[3], fn, last);
        som.dumpOnline(out);
        out.write((const char *)last.data(), (std::streamsize)(last.size() * 8));
        out.write((const char *)&mse, 4);
    }
    return out.good() ? 0 : 4;
}

#else   // no Eigen on the include path

#include <cstdio>
int main()
{
    std::fprintf(stderr, "eigen_crosscheck: built without <Eigen/Dense> -- Eigen absent, nothing to cross-check\n");
    return 77;
}

#endif
