// vsom_api.hpp -- host-side C++ interface of the MI355X VSOM build.
//
// One header declares the types a caller of the reference's libsom uses on the training hot path
// (ColumnSpec, RowData, IDataLoader, DataSet, SomIndex, UMatrix, Transformation, Som), with the
// reference's names, member signatures and error behaviour, so that code written against
// include/SOM.hpp & co. recompiles against this build.  The small headers SOM.hpp,
// Transformation.hpp, DataSet.hpp, IDataLoader.hpp, SomIndex.hpp, UMatrix.hpp, ColumnSpec.hpp next
// to this file only forward to it.  Every hot-path member of Som is a thin call into the C ABI of
// libvsom_hip.so (include/vsom_hip.h); nothing here computes training arithmetic on the CPU.
#pragma once
#include "vsom_dense.hpp"
#include <cstdint>

// ===== ColumnSpec ============================================================================
// Mirror of the reference's include/ColumnSpec.hpp:5-16 (loader-side value type).
#include <string>

struct ColumnSpec {
    ColumnSpec(const std::string name, const float weight, const int isBinary)
        : name{name}, weight{weight}, isBinary{isBinary} {}
    const std::string name;
    const float weight;
    const int isBinary;
};

// ===== IDataLoader ===========================================================================
// Mirror of the reference's include/IDataLoader.hpp:11-48 (abstract chunked loader).  The SQLite
// and MNIST loaders of the reference are out of scope; ArrayDataLoader is the in-memory loader the
// tests and examples use.

#include <optional>
#include <string>
#include <vector>

struct RowData {
    Eigen::VectorXf values;
    std::vector<int> valid;
};

class IDataLoader {
public:
    virtual ~IDataLoader() = default;
    IDataLoader(std::optional<size_t> maxLoadCount = std::nullopt)
        : m_maxLoadCount{maxLoadCount}, m_currentIndex{0} {}
    virtual size_t load() = 0;
    virtual std::vector<RowData> getPreview(size_t count) = 0;
    virtual bool open(const char *path) = 0;
    virtual std::vector<std::string> findAllColumns() = 0;
    std::vector<RowData> data;

    virtual void setColumnSpec(const std::vector<ColumnSpec> columnSpec) noexcept = 0;
    virtual const std::vector<ColumnSpec> getColumnSpec() noexcept = 0;
    virtual float getWeight(size_t index) = 0;
    virtual const std::vector<float> getWeights() const noexcept = 0;
    virtual const std::vector<int> &getBinary() const noexcept = 0;
    virtual const std::vector<int> &getContinuous() const noexcept = 0;
    virtual std::string getName(size_t index) const noexcept = 0;
    virtual const std::vector<std::string> getNames() const noexcept = 0;
    virtual size_t getDepth() const noexcept = 0;
    virtual bool isAtStartOfDataStream() const noexcept = 0;

    // [MI355X build] optional fast path of DataSet::loadNextDataFromStream (the step right before the hot path:
    // the reference reloads every chunk every epoch, Som.cpp:737,1157).  peekFlat: this loader can write a chunk
    // straight into a caller's buffer, and `rows` is what the NEXT load() would return.  loadFlat: write those rows
    // (row-major, getDepth() floats each; every value valid) into dst, advance the stream exactly as load() does and
    // leave `data` empty.  A loader that overrides neither keeps the reference's contract: load() fills `data`.
    virtual bool peekFlat(size_t &rows) { (void)rows; return false; }
    virtual size_t loadFlat(float *dst, size_t rows) { (void)dst; (void)rows; return 0; }

protected:
    std::optional<size_t> m_maxLoadCount;
    size_t m_currentIndex;
};

// In-memory loader: rows x depth floats, load() yields the next <= maxLoadCount rows and wraps
// to the start after the last chunk (the streaming contract of SqliteDataLoader.cpp:465-479).
class ArrayDataLoader : public IDataLoader {
public:
    ArrayDataLoader(const float *rows, size_t nrows, size_t depth,
                    std::optional<size_t> maxLoadCount = std::nullopt);
    size_t load() override;
    bool peekFlat(size_t &rows) override;
    size_t loadFlat(float *dst, size_t rows) override;
    std::vector<RowData> getPreview(size_t count) override;
    bool open(const char *) override { return true; }
    std::vector<std::string> findAllColumns() override { return getNames(); }
    void setColumnSpec(const std::vector<ColumnSpec> columnSpec) noexcept override;
    const std::vector<ColumnSpec> getColumnSpec() noexcept override;
    float getWeight(size_t index) override { return m_weights[index]; }
    const std::vector<float> getWeights() const noexcept override { return m_weights; }
    const std::vector<int> &getBinary() const noexcept override { return m_binary; }
    const std::vector<int> &getContinuous() const noexcept override { return m_continuous; }
    std::string getName(size_t index) const noexcept override { return m_names[index]; }
    const std::vector<std::string> getNames() const noexcept override { return m_names; }
    size_t getDepth() const noexcept override { return m_depth; }
    bool isAtStartOfDataStream() const noexcept override { return m_currentIndex == 0; }

private:
    std::vector<float> m_rows;
    size_t m_nrows, m_depth;
    std::vector<float> m_weights;
    std::vector<int> m_binary, m_continuous;
    std::vector<std::string> m_names;
};

// ----- MnistDataLoader -------------------------------------------------------------------------
// Mirror of the reference's include/MnistDataLoader.hpp:9-51 / src/MnistDataLoader.cpp:9-134.
// open(folder) remembers the folder; load() reads the next <= maxLoadCount images of
// `folder/train-images-idx3-ubyte` + `train-labels-idx1-ubyte` (IDX, big-endian headers, magic
// 0x803 / 0x801): 784 raw pixel values 0..255 as float followed by the one-hot label (10 values),
// depth 794, valid = 1 (MnistDataLoader.cpp:61-81).  The stream position advances by the rows read
// and returns to 0 after an empty read or a read of >= 60000 rows (:53-55).  Own IDX parser (the
// reference delegates to extern/mnistReader).
class MnistDataLoader : public IDataLoader {
public:
    MnistDataLoader(std::optional<size_t> maxLoadCount = std::nullopt, bool verbose = false);
    size_t load() override;
    bool peekFlat(size_t &rows) override;
    size_t loadFlat(float *dst, size_t rows) override;
    std::vector<RowData> getPreview(size_t count) override;
    bool open(const char *path) override;
    std::vector<std::string> findAllColumns() override { return _names; }
    void setColumnSpec(const std::vector<ColumnSpec>) noexcept override {}
    const std::vector<ColumnSpec> getColumnSpec() noexcept override;
    float getWeight(size_t index) override { return _weights.at(index); }
    const std::vector<float> getWeights() const noexcept override { return _weights; }
    const std::vector<int> &getBinary() const noexcept override { return _isBinary; }
    const std::vector<int> &getContinuous() const noexcept override { return _isContinuous; }
    std::string getName(size_t index) const noexcept override { return _names.at(index); }
    const std::vector<std::string> getNames() const noexcept override { return _names; }
    size_t getDepth() const noexcept override { return _names.size(); }
    bool isAtStartOfDataStream() const noexcept override { return m_currentIndex == 0; }

protected:
    std::vector<float> _weights;
    std::vector<int> _isBinary, _isContinuous;
    std::vector<std::string> _names;
    std::string _filePath;
    bool _verbose;
    mutable std::vector<unsigned char> _img, _lab;   // file bytes, kept between loads
    std::vector<RowData> readRows(size_t skip, size_t limit) const;
    size_t rowsAvailable(size_t skip, size_t limit) const;     // rows readRows(skip, limit) would return
    void convertRows(size_t skip, size_t nrows, float *flat, std::vector<RowData> *rows) const;
};

// ----- SqliteDataLoader ------------------------------------------------------------------------
// Mirror of the reference's include/SqliteDataLoader.hpp:11-103 / src/SqliteDataLoader.cpp for the
// calls the training drivers and tests/performance/perf_tests.cpp:77-84 make: open, setTable,
// setColumnSpec / column-spec file, findAllTables, findAllColumns, load (Id-range chunk query
// `SELECT <cols>,Id FROM <table> WHERE Id>=? AND Id<=?`, SqliteDataLoader.cpp:481-548), getPreview.
// The SQLite C library is bound at run time (dlopen of libsqlite3.so.0): the build image has the
// shared object but no header, and the reference's vendored extern/sqlite/sqlite3.c is absent.
struct sqlite3;
class SqliteDataLoader : public IDataLoader {
public:
    SqliteDataLoader(const std::string &specFilePath, std::optional<size_t> maxLoadCount = std::nullopt,
                     bool verbose = false);
    SqliteDataLoader(bool db, const std::string &dbPath, std::optional<size_t> maxLoadCount = std::nullopt);
    ~SqliteDataLoader() override;
    SqliteDataLoader(const SqliteDataLoader &) = delete;
    SqliteDataLoader &operator=(const SqliteDataLoader &) = delete;
    void setTable(const std::string &name) noexcept { _tableName = name; }
    void setColumnSpec(const std::vector<ColumnSpec> columnSpec) noexcept override;
    const std::vector<ColumnSpec> getColumnSpec() noexcept override { return _columnSpec; }
    size_t load() override;
    std::vector<RowData> getPreview(size_t count) override;
    bool open(const char *dbPath) override;
    bool open() { return open(_dbPath.c_str()); }
    std::vector<std::string> findAllColumns() override;
    std::vector<std::string> findAllTables();
    const std::vector<float> getWeights() const noexcept override;
    float getWeight(size_t index) override { return _columnSpec.at(index).weight; }
    const std::vector<int> &getBinary() const noexcept override { return _isBinary; }
    const std::vector<int> &getContinuous() const noexcept override { return _isContinuous; }
    std::string getName(size_t index) const noexcept override { return _columnSpec.at(index).name; }
    const std::vector<std::string> getNames() const noexcept override;
    size_t getDepth() const noexcept override { return vectorLength; }
    bool isAtStartOfDataStream() const noexcept override { return !currentLoadId.has_value(); }

protected:
    std::vector<ColumnSpec> _columnSpec;
    std::vector<std::string> _tableNames, _columnNames;
    std::vector<int> _isBinary, _isContinuous;
    std::string _tableName, _dbPath;
    sqlite3 *db = nullptr;
    bool _verbose = false, hasOpenDatabase = false;
    size_t vectorLength = 0;
    std::optional<long long> currentLoadId;
    void loadColumnSpecData(const std::string &path);
    std::vector<std::string> queryStrings(const std::string &sql);
    long long queryInteger(const std::string &sql);
    // rows with Id in [startId or MIN(Id), ...], at most maxCount ids wide; returns {rows, next id, MAX(Id)}
    struct Fetch { std::vector<RowData> rows; std::optional<long long> nextId; long long maxId; };
    Fetch fetch(std::optional<long long> startId, std::optional<size_t> maxCount);
};

// ===== DataSet ===============================================================================
// Mirror of the reference's include/DataSet.hpp:10-62 (chunk container consumed by the hot path).

#include <string>
#include <vector>

class DataSet {
protected:
    struct DataRow {
        Eigen::VectorXf *data;
        std::vector<int> *valid;
        size_t *lastBMU;
    };
    // [MI355X build] built lazily by ensureRows() (hence mutable): see vsom_host.cpp
    mutable std::vector<DataRow> allData;
    mutable std::vector<Eigen::VectorXf> data;
    mutable std::vector<std::vector<int>> valid;
    std::vector<size_t> index;
    mutable std::vector<size_t> lastBMU;
    IDataLoader &_loader;
    size_t depth, n, loadedNumberOfChunks;
    bool _verbose;

public:
    DataSet(IDataLoader &dataLoader, bool verbose = false)
        : _loader{dataLoader}, depth{}, n{}, loadedNumberOfChunks{0}, _verbose{verbose} {}
    ~DataSet() = default;   // (copyable like the reference's: see the copy constructor below)
    const std::vector<DataRow> getAll() const;
    std::vector<DataRow> getAll();
    std::vector<Eigen::VectorXf> getPreviewData(size_t count) const;
    Eigen::VectorXf getData(size_t index) const;
    const Eigen::VectorXi getValidity(size_t index) const;
    const Eigen::ArrayXi getBinary() const;
    const Eigen::ArrayXi getContinuous() const;
    const Eigen::VectorXf getWeights() const;
    float getWeight(size_t index);
    const std::vector<std::string> getNames() const noexcept;
    std::string getName(size_t) const;
    const std::vector<size_t> &getLastBMU() const noexcept;
    size_t &getLastBMU(size_t);
    size_t size() const;
    void addVector(Eigen::VectorXf);
    void display() const;
    void loadNextDataFromStream();
    size_t vectorLength() const;
    bool hasReadWholeDataStream() const noexcept;
    void resetStreamLoadPosition() noexcept;
    void shuffle();

    // [MI355X build] contiguous B x depth row-major copy of the current chunk in PINNED host memory
    // (two buffers used alternately, so the asynchronous host->device copy of one chunk can still be
    // running while the next chunk is being loaded into the other)
    const float *contiguous() const noexcept { return m_flat[m_cur].p; }
    DataSet(const DataSet &other);
    DataSet &operator=(const DataSet &) = delete;

private:
    struct Pinned {
        float *p = nullptr;
        size_t cap = 0;
        bool pinned = false;
        Pinned() = default;
        Pinned(const Pinned &) = delete;
        Pinned &operator=(const Pinned &) = delete;
        ~Pinned();
        void reserve(size_t n);
    };
    Pinned m_flat[2];
    int m_cur = 0;
    mutable bool m_rowsBuilt = true;
    bool m_fromFlat = false;      // the current chunk came through IDataLoader::loadFlat (the loader's `data` is empty)
    void ensureRows() const;
};

// ===== SomIndex ==============================================================================
// Mirror of the reference's include/SomIndex.hpp:7-27.
#include <stddef.h>

class Som;

class SomIndex {
protected:
    size_t x, y;

public:
    SomIndex(size_t x = 0, size_t y = 0) noexcept;
    SomIndex(const Som &map, size_t index) noexcept;   // y = (index - index % W) / H  (SomIndex.cpp:13-18)
    ~SomIndex() = default;
    size_t getSomIndex(const Som &som);
    size_t getX() const noexcept;
    size_t getY() const noexcept;
    void setX(size_t index) noexcept;
    void setY(size_t index) noexcept;
    bool operator==(const SomIndex &other) const { return x == other.x && y == other.y; }
};

// ===== UMatrix ===============================================================================
// Mirror of the reference's include/UMatrix.hpp:6-23 (value type; the U-matrix computation itself
// is outside the hot path).
#include <vector>

class UMatrix {
private:
    std::vector<double> data;
    size_t width, height;

public:
    UMatrix(std::vector<double> Data, size_t Width, size_t Height) : data{Data}, width{Width}, height{Height} {}
    double getValueAtIndex(size_t Width, size_t Height) const { return data[Height * width + Width]; }
    double getValueAtIndex(SomIndex Index) const { return data[Index.getY() * width + Index.getX()]; }
    const std::vector<double> &getData() const noexcept { return data; }
    size_t getWidth() const noexcept { return width; }
    size_t getHeight() const noexcept { return height; }
};

// ===== Transformation ========================================================================
// Mirror of the reference's include/Transformation.hpp:10-41: same public members, same three
// factories, aggregate-initialisable with designated initialisers (tests/test1.cpp:56-84).
//
// [MI355X build] the std::function hooks cannot run on the device.  The three built-in
// transformations are recognised by the *type* of the callable stored in Comparer/Stepper
// (vsom::*Comparer / vsom::*Stepper below) and run as HIP kernels -- they have no host training path.
// Any other callable (a caller's own lambdas, tests/test1.cpp:56-84) makes the Som keep its state on the
// host and run distance / search / batch epoch / online step there with the caller's hooks
// (src/vsom_custom.cpp); the consumers outside training (U-matrix, evaluate, ...) then throw.  The
// built-in functors also work on the host, so code that calls transform.Comparer(...) directly keeps working.

#include <functional>
#include <sstream>
#include <string>
#include <vector>

namespace vsom {
using T = Eigen::VectorXf;
struct StandardComparer { T operator()(const T &value, const T &model, const T &dispersion, const T &valueWeight) const; };
struct StandardStepper { T operator()(const T &value, const T &model, const T &valueWeight) const; };
struct MedianStepper { T operator()(const T &value, const T &model, const T &valueWeight) const; };
struct ClrComparer { T operator()(const T &value, const T &model, const T &dispersion, const T &valueWeight) const; };
struct ClrStepper { T operator()(const T &value, const T &model, const T &valueWeight) const; };
struct IdentityLength { size_t operator()(size_t vectorLength) const { return vectorLength; } };
struct ClrLength { size_t operator()(size_t vectorLength) const { return vectorLength * (vectorLength - 1u); } };
enum Kind { Standard = 0, Median = 1, Clr = 2, Custom = -1 };
}   // namespace vsom

struct Transformation {
    using T = Eigen::VectorXf;
    std::function<T(const T &value, const T &model, const T &dispersion, const T &valueWeight)> Comparer{vsom::StandardComparer{}};
    std::function<T(const T &value, const T &model, const T &valueWeight)> Stepper{vsom::StandardStepper{}};
    std::vector<std::string> names{};
    std::function<std::vector<std::string>(const T &model)> Displayer{[&names = names](const T &) { return names; }};
    std::function<size_t(size_t vectorLength)> Length{vsom::IdentityLength{}};
    std::string Name{"Standard transformation"};

    static Transformation Standard(const std::vector<std::string> &columnNames);
    static Transformation StandardMedianEstimator(const std::vector<std::string> &columnNames);
    static Transformation CombinatorialLinearRegression(const std::vector<std::string> &columnNames);

    // [MI355X build] which device kernels implement this transformation (vsom::Custom = none)
    int kind() const noexcept;
};

// ===== Som ===================================================================================
// Mirror of `class Som` (reference include/SOM.hpp:39-189).  State lives in HBM inside a vsom_ctx;
// the getters copy it back on demand (they return by value in the reference too).
#include <atomic>
#include <mutex>

#define VERSION 1.00

// argv positions of the reference's CLI (include/SOM.hpp:17-35; apps/main.cpp indexes argv with them)
#define ARG_SETTING 2

#define ARG_DB_FILE 3
#define ARG_SOM_FILE 4

// Training parameters
#define ARG_SOM_HEIGHT 5
#define ARG_SOM_WIDTH 6
#define ARG_SOM_ETA0 7
#define ARG_SOM_ETA_DEC 8
#define ARG_SOM_SIGMA0 9
#define ARG_SOM_SIGMA_DEC 10
#define ARG_SOM_EPOCHS 11
#define ARG_SOM_INIT_SIGMA 12
#define ARG_SOM_WEIGHT_DECAY_FUNCTION 13

// Measuring parameters
#define ARG_ALLOWED_STD_DEV 6
#define ARG_MIN_BMU_HITS 5

#define SIGMA_SWITCH_TO_LOCAL 1

struct vsom_ctx;
struct vsom_group;

class Som {
protected:
    struct TrainingReturnValue {
        SomIndex bmu;
        Eigen::VectorXf residual;
        float distanceError;
    };
    struct Metrics {
        std::vector<float> MeanSquaredError;
        std::vector<float> DistanceError;
        Metrics() : MeanSquaredError{}, DistanceError{} {}
        Metrics(size_t size) : MeanSquaredError(size), DistanceError(size) {}
    };
    Transformation transform;
    Metrics metrics;
    std::vector<double> uMatrix;
    std::atomic<bool> _isTraining;
    size_t height, width, depth;

    void Construct(size_t inWidth, size_t inHeight, size_t inDepth, std::vector<std::string> names);

public:
    enum class WeigthDecayFunction { Exponential, InverseProportional, BatchMap };
    std::mutex metricsMutex;

    Som(size_t width, size_t height, DataSet dataset, Transformation transformation = Transformation{});
    Som(size_t width, size_t height, size_t depth, Transformation transformation = Transformation{});
    Som(const char *filename);
    Som(const Som &som);
    Som &operator=(const Som &other);
    ~Som();

    // ---- training hot path (src/Som.cpp:716-947, 1113-1192) -> libvsom_hip.so ----------------
    void train(DataSet &data, size_t numberOfEpochs, double eta0, double etaDecay, double sigma0,
               double sigmaDecay, WeigthDecayFunction weightDecayFunction, bool updateUMatrixAfterEpoch = false);
    void trainBasicSom(DataSet &data, size_t numberOfEpochs, double eta0, double etaDecay, double sigma0,
                       double sigmaDecay, WeigthDecayFunction weightDecayFunction, bool updateUMatrixAfterEpoch = false);
    void trainBatchSom(DataSet &data, size_t numberOfEpochs, double sigma0, double sigmaDecay,
                       bool updateUMatrixAfterEpoch = false);
    float trainBatchSomEpoch(DataSet &data, double currentSigma, bool isFirst);
    TrainingReturnValue trainSingle(const Eigen::VectorXf &v, const Eigen::VectorXf &valid,
                                    const Eigen::VectorXf &weights, const double eta, const double sigma,
                                    size_t &lastBMU, const WeigthDecayFunction weightDecayFunction);
    SomIndex findBmu(const Eigen::VectorXf &v) const;
    SomIndex findBmu(const Eigen::VectorXf &v, const Eigen::VectorXf &valid, const Eigen::VectorXf &weights) const;
    SomIndex findLocalBmu(const Eigen::VectorXf &v, const Eigen::VectorXf &valid, const size_t &lastBMUref,
                          const Eigen::VectorXf &weights) const;
    double euclidianWeightedDist(const SomIndex &pos, const Eigen::VectorXf &v, const Eigen::VectorXf &valid,
                                 const Eigen::VectorXf &weights) const;
    double euclidianWeightedDist(const size_t &pos, const Eigen::VectorXf &v, const Eigen::VectorXf &valid,
                                 const Eigen::VectorXf &weights) const;
    double static calculateNeighbourhoodWeight(const size_t &currentX, const size_t &currentY, const size_t &bmuX,
                                               const size_t &bmuY, const double &currentSigma);

    // ---- consumers of the search outside the training loop: distances and
    //      (restricted) BMU searches on the device, scalar post-processing on the host ------------
    double evaluate(const DataSet &dataset) const;
    int measureSimilarity(const DataSet *dataset, int numberOfSigmas, size_t minBmuHits) const;
    int autoEncoder(const DataSet *dataset, size_t minBmuHits) const;
    size_t variationalAutoEncoder(const DataSet *dataset, size_t minBmuHits) const;
    SomIndex findRestrictedBmu(const Eigen::VectorXf &v, const Eigen::VectorXf &valid, const size_t minBmuHits,
                               const Eigen::VectorXf &weights) const;
    std::vector<double> findRestrictedBmd(const Eigen::VectorXf &v, const Eigen::VectorXf &valid, size_t minBmuHits,
                                          const Eigen::VectorXf &weights) const;
    double euclidianWeightedDistRaw(const size_t &pos, const Eigen::VectorXf &v, const Eigen::VectorXf &valid,
                                    const Eigen::VectorXf &weights) const;
    void updateUMatrix(const Eigen::VectorXf &weights);

    // ---- accessors / bookkeeping (src/Som.cpp:85-110, 159-281, 977-997, 1189-1206) -------------
    void display() const;
    void displayUMatrix() const;
    UMatrix getUMatrix() const noexcept;
    Eigen::VectorXf getWeigthMap() const noexcept;
    std::vector<size_t> getBmuHits() const noexcept;
    size_t getHeight() const noexcept;
    size_t getWidth() const noexcept;
    size_t getDepth() const noexcept;
    size_t getIndex(SomIndex index) const noexcept;
    Eigen::VectorXf getNeuron(SomIndex index) const noexcept;
    Eigen::VectorXf getNeuron(size_t index) const noexcept;
    Eigen::VectorXf getSigmaNeuron(SomIndex index) const noexcept;
    Eigen::VectorXf getSigmaNeuron(size_t index) const noexcept;
    std::vector<std::string> getNeuronStrings(SomIndex index) const noexcept;
    std::vector<std::string> getSigmaNeuronStrings(SomIndex index) const noexcept;
    float getMaxValueOfFeature(size_t modelVectorIndex) const;
    float getMinValueOfFeature(size_t modelVectorIndex) const;
    float getMaxSigmaOfFeature(size_t modelVectorIndex) const;
    float getMinSigmaOfFeature(size_t modelVectorIndex) const;
    Metrics getMetrics() const noexcept;
    bool isTraining() const noexcept;
    bool isCompatibleWithData(DataSet &data) const noexcept;
    void randomInitialize(int seed, float sigma);
    void addBmu(SomIndex position);
    void save(const char *filename) const;   // Octave text format of the reference (Som.cpp:1209-1294)
    void load(const char *filename);         // that format (Som.cpp:1343-1597), or saveBinary's
    // Som.cpp:1296-1341: sets width / height from the file's "# columns:" / "# rows:" lines and returns a
    // vector of the file's model-vector length (length 1 when the file names none).  [MI355X build] when
    // the dimensions change, the device state is rebuilt zeroed at the new size (the reference leaves its
    // `map` vectors at the old size, which the next access overruns).
    Eigen::VectorXf getSizeFromFile(const char *filename);
    void saveBinary(const char *filename) const;   // [MI355X build] lossless, incl. SMap

    // ---- [MI355X build] bulk state access (row-major N x depth) and device selection -----------
    void setState(const float *map, const float *sigma, const float *S, const float *weight, const uint64_t *hits);
    void getState(float *map, float *sigma, float *S, float *weight, uint64_t *hits) const;
    static void setDefaultDevice(int device);
    // devices the batch-map training runs on (opt-in; default: the default device alone): {} = the list in
    // VSOM_DEVICES or the default device, one entry = single GPU, several = sample/node-sharded epochs through
    // vsom_group_* (vsom_host.cpp, training_devices)
    static void setDevices(const std::vector<int> &devices);
    // arithmetic of the Standard update chains of Soms created afterwards: false = strict, bit-identical to
    // the reference (default); true = contracted (fused multiply-adds, results within 1e-5: vsom_hip.h)
    static void setContractedArithmetic(bool on);
    // any vsom_update_mode of include/vsom_hip.h (0 strict, 1 contracted, 2 only the sigma^2 accumulation
    // contracted: map and BMUs stay bit-identical over whole schedules)
    static void setUpdateArithmetic(int mode);
    vsom_ctx *context() const noexcept { return ctx; }
    vsom_group *group() const noexcept { return grp; }

private:
    vsom_ctx *ctx = nullptr;          // the context searches, getters and the online path use (member 0 of grp)
    vsom_group *grp = nullptr;        // set when the Som trains on more than one GPU; owns ctx then
    bool replicasStale = false;       // member 0 was trained alone (online path): the others lag behind
    void destroyContext();
    void copyStateFrom(const Som &other);
    void joinGroup() const;
    void syncReplicas();
    size_t inLen = 0;                 // J: sample length (depth = transform.Length(J))
    mutable bool hostStale = true;    // host mirrors below are out of date
    mutable std::vector<float> hMap, hSigma, hWeight;
    mutable std::vector<uint64_t> hHits;
    // host execution for user-supplied hooks (kind() == vsom::Custom; src/vsom_custom.cpp): the state then
    // lives in hMap / hSigma / hS / hWeight / hHits and ctx stays null
    mutable std::vector<float> hS;
    void hostEnsure() const;
    double hostDist(size_t pos, const Eigen::VectorXf &v, const Eigen::VectorXf &valid, const Eigen::VectorXf &weights) const;
    size_t hostFindBmu(const Eigen::VectorXf &v, const Eigen::VectorXf &valid, const Eigen::VectorXf &weights) const;
    size_t hostFindLocalBmu(const Eigen::VectorXf &v, const Eigen::VectorXf &valid, size_t start, const Eigen::VectorXf &weights) const;
    size_t hostFindRestrictedBmu(const Eigen::VectorXf &v, const Eigen::VectorXf &valid, size_t minBmuHits, const Eigen::VectorXf &weights) const;
    double hostDistRaw(size_t pos, const Eigen::VectorXf &v, const Eigen::VectorXf &valid, const Eigen::VectorXf &weights) const;
    float hostBatchEpoch(DataSet &dataset, double currentSigma, bool isFirst);
    void hostTrainBatchSom(DataSet &data, size_t numberOfEpochs, double sigma0, double sigmaDecay, bool updateUMatrixAfterEpoch);
    TrainingReturnValue hostTrainSingle(const Eigen::VectorXf &v, const Eigen::VectorXf &valid, const Eigen::VectorXf &weights,
                                        double eta, double sigma, size_t &lastBMU, WeigthDecayFunction fn);
    void hostTrainBasicSom(DataSet &data, size_t numberOfEpochs, double eta0, double etaDecay, double sigma0, double sigmaDecay,
                           WeigthDecayFunction fn, bool updateUMatrixAfterEpoch);
    void createContext();
    void requireDevicePath(const char *what) const;
    void refreshHost() const;
    void stageOne(const Eigen::VectorXf &v) const;
};
