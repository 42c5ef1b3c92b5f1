// vsom_loaders.cpp -- the reference's two file-backed IDataLoader implementations for the MI355X
// build's host library (the step immediately before the hot path).
//
//   MnistDataLoader   reference include/MnistDataLoader.hpp, src/MnistDataLoader.cpp (which delegates
//                     the file format to extern/mnistReader); here with its own IDX parser.
//   SqliteDataLoader  reference include/SqliteDataLoader.hpp, src/SqliteDataLoader.cpp; the SQLite C
//                     library is bound at run time with dlopen (no header / no vendored sqlite3.c in
//                     the build image).
// Loader semantics (chunk sizes, stream position, column order, value conversion) follow the
// reference lines cited at each function; the bytes they hand to DataSet are what the reference's
// loaders would hand over.
#include "vsom_api.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <thread>

// ---------------------------------------------------------------------------------------------
// MnistDataLoader
// ---------------------------------------------------------------------------------------------
MnistDataLoader::MnistDataLoader(std::optional<size_t> maxLoadCount, bool verbose)
    : IDataLoader{maxLoadCount}, _weights(28 * 28 + 10, 1.0f), _isBinary(28 * 28 + 10, 0),
      _isContinuous(28 * 28 + 10, 1), _names{}, _filePath{}, _verbose{verbose}
{
    // MnistDataLoader.cpp:115-134: "<x>x<y>" for the pixels, "label:<k>" for the one-hot label
    _names.reserve(28 * 28 + 10);
    for (size_t i = 0; i < 28 * 28; ++i)
        _names.push_back(std::to_string(i % 28) + "x" + std::to_string(i / 28));
    for (size_t k = 0; k < 10; ++k)
        _names.push_back("label:" + std::to_string(k));
}

bool MnistDataLoader::open(const char *path)
{
    _filePath = path;   // MnistDataLoader.cpp:9-14: remembers the folder, nothing is read yet
    _img.clear();
    _lab.clear();
    return true;
}

const std::vector<ColumnSpec> MnistDataLoader::getColumnSpec() noexcept
{
    std::vector<ColumnSpec> out;
    for (size_t i = 0; i < _names.size(); ++i)
        out.emplace_back(_names[i], _weights[i], _isBinary[i]);
    return out;
}

namespace {

uint32_t be32(const unsigned char *p)
{
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

// whole IDX file with a checked magic number and length (mnist_reader_common.hpp:36-76)
std::vector<unsigned char> read_idx(const std::string &path, uint32_t magic)
{
    std::ifstream f(path, std::ios::in | std::ios::binary | std::ios::ate);
    if (!f) {
        std::cout << "Error opening file" << std::endl;
        return {};
    }
    const std::streamsize size = f.tellg();
    std::vector<unsigned char> buf((size_t)std::max<std::streamsize>(size, 0));
    f.seekg(0, std::ios::beg);
    if (size > 0)
        f.read(reinterpret_cast<char *>(buf.data()), size);
    if (buf.size() < 8 || be32(buf.data()) != magic) {
        std::cout << "Invalid magic number, probably not a MNIST file" << std::endl;
        return {};
    }
    const size_t count = be32(buf.data() + 4);
    if (magic == 0x803) {
        if (buf.size() < 16 || buf.size() < count * be32(buf.data() + 8) * be32(buf.data() + 12) + 16) {
            std::cout << "The file is not large enough to hold all the data, probably corrupted" << std::endl;
            return {};
        }
    } else if (buf.size() < count + 8) {
        std::cout << "The file is not large enough to hold all the data, probably corrupted" << std::endl;
        return {};
    }
    return buf;
}

// rows to read from a file of `count` rows (mnist_reader.hpp:69-89)
size_t amount_to_read(size_t limit, size_t skip, size_t count)
{
    if (limit > 0 && count > limit + skip)
        return limit;
    if (limit > 0)
        return count > skip ? count - skip : 0;
    if (skip >= count)
        return 0;
    return count;   // no limit: the reference returns the full count even after a skip
}

}   // namespace

// rows readRows(skip, limit) returns; reads the two files on first use (the reference re-reads both on every
// load(); the bytes are kept after the first read here -- open() drops them -- which changes nothing unless the files
// are rewritten during training)
size_t MnistDataLoader::rowsAvailable(size_t skip, size_t limit) const
{
    if (_img.empty() || _lab.empty()) {
        _img = read_idx(_filePath + "/train-images-idx3-ubyte", 0x803);
        _lab = read_idx(_filePath + "/train-labels-idx1-ubyte", 0x801);
    }
    if (_img.empty() || _lab.empty())
        return 0;
    const size_t icount = be32(_img.data() + 4), lcount = be32(_lab.data() + 4);
    size_t ni = amount_to_read(limit, skip, icount), nl = amount_to_read(limit, skip, lcount);
    // never past the buffers (a no-limit read after a skip is clipped; the reference would over-read)
    ni = std::min(ni, icount > skip ? icount - skip : 0);
    nl = std::min(nl, lcount > skip ? lcount - skip : 0);
    return std::min(ni, nl);   // std::transform over images, zipped with labels (:61-64)
}

// rows [skip, skip + nrows) as 784 raw pixel values + the one-hot label, into a contiguous buffer (`flat`) or into
// the reference's per-row containers (`rows`).  Rows are independent: a few host threads convert a chunk (4096 x 794
// values is several milliseconds on one core -- as long as the device step it has to hide behind)
void MnistDataLoader::convertRows(size_t skip, size_t nrows, float *flat, std::vector<RowData> *rows) const
{
    const size_t px = be32(_img.data() + 8) * be32(_img.data() + 12), depth = px + 10;
    auto convert = [&](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; ++r) {
            float *dst;
            if (rows) {
                RowData &row = (*rows)[r];
                row.values = Eigen::VectorXf((Eigen::Index)depth);
                row.valid.assign(depth, 1);
                dst = row.values.data();
            } else {
                dst = flat + r * depth;
            }
            const unsigned char *src = _img.data() + 16 + (skip + r) * px;
            for (size_t d = 0; d < px; ++d)
                dst[d] = (float)src[d];                               // raw 0..255, un-normalised (:73-75)
            const unsigned label = _lab[8 + skip + r];
            for (size_t k = 0; k < 10; ++k)
                dst[px + k] = (label == k) ? 1.0f : 0.0f;             // one-hot label (:66-71)
        }
    };
    const size_t nthreads = nrows >= 512 ? std::min<size_t>(8, std::max(1u, std::thread::hardware_concurrency())) : 1;
    if (nthreads <= 1) {
        convert(0, nrows);
    } else {
        std::vector<std::thread> pool;
        const size_t per = (nrows + nthreads - 1) / nthreads;
        for (size_t t = 0; t < nthreads; ++t) {
            const size_t r0 = t * per, r1 = std::min(nrows, r0 + per);
            if (r0 < r1)
                pool.emplace_back(convert, r0, r1);
        }
        for (auto &th : pool)
            th.join();
    }
}

std::vector<RowData> MnistDataLoader::readRows(size_t skip, size_t limit) const
{
    std::vector<RowData> out;
    const size_t nrows = rowsAvailable(skip, limit);
    if (nrows == 0)
        return out;
    out.resize(nrows);
    convertRows(skip, nrows, nullptr, &out);
    return out;
}

bool MnistDataLoader::peekFlat(size_t &rows)
{
    rows = rowsAvailable(m_currentIndex, m_maxLoadCount.value_or(0));
    // The flat path writes rows x cols + 10 values per row into a buffer the caller sized with getDepth() (794).  An IDX
    // file whose images are not 28 x 28 does not fit that: no flat path -- load() clips every row to the depth instead.
    if (rows > 0 && (size_t)be32(_img.data() + 8) * be32(_img.data() + 12) + 10 != getDepth())
        return false;
    return true;
}

size_t MnistDataLoader::loadFlat(float *dst, size_t rows)
{
    const size_t n = std::min(rows, rowsAvailable(m_currentIndex, m_maxLoadCount.value_or(0)));
    data.clear();
    if (n)
        convertRows(m_currentIndex, n, dst, nullptr);
    if (n == 0 || n >= 60000)     // MnistDataLoader.cpp:53-55
        m_currentIndex = 0;
    else
        m_currentIndex += n;
    return n;
}

size_t MnistDataLoader::load()
{
    const size_t limit = m_maxLoadCount.value_or(0);
    data = readRows(m_currentIndex, limit);
    const size_t n = data.size();
    if (n == 0 || n >= 60000)     // MnistDataLoader.cpp:53-55
        m_currentIndex = 0;
    else
        m_currentIndex += n;
    return n;
}

std::vector<RowData> MnistDataLoader::getPreview(size_t count)
{
    return readRows(0, count);    // MnistDataLoader.cpp:16-45
}

// ---------------------------------------------------------------------------------------------
// SqliteDataLoader
// ---------------------------------------------------------------------------------------------
namespace {

struct sqlite3_stmt;
constexpr int kSqliteOk = 0, kSqliteRow = 100, kSqliteOpenReadonly = 1;

// the handful of SQLite entry points the loader needs, resolved once from libsqlite3.so.0
struct SqliteApi {
    int (*open_v2)(const char *, sqlite3 **, int, const char *) = nullptr;
    int (*close)(sqlite3 *) = nullptr;
    const char *(*errmsg)(sqlite3 *) = nullptr;
    int (*prepare_v2)(sqlite3 *, const char *, int, sqlite3_stmt **, const char **) = nullptr;
    int (*bind_int64)(sqlite3_stmt *, int, long long) = nullptr;
    int (*step)(sqlite3_stmt *) = nullptr;
    double (*column_double)(sqlite3_stmt *, int) = nullptr;
    long long (*column_int64)(sqlite3_stmt *, int) = nullptr;
    const unsigned char *(*column_text)(sqlite3_stmt *, int) = nullptr;
    int (*finalize)(sqlite3_stmt *) = nullptr;
    bool ok = false;
};

const SqliteApi &sqlite_api()
{
    static const SqliteApi api = [] {
        SqliteApi a;
        void *h = nullptr;
        for (const char *name : {"libsqlite3.so.0", "libsqlite3.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h)
                break;
        }
        if (!h)
            return a;
        auto sym = [&](const char *n) { return dlsym(h, n); };
        a.open_v2 = reinterpret_cast<decltype(a.open_v2)>(sym("sqlite3_open_v2"));
        a.close = reinterpret_cast<decltype(a.close)>(sym("sqlite3_close"));
        a.errmsg = reinterpret_cast<decltype(a.errmsg)>(sym("sqlite3_errmsg"));
        a.prepare_v2 = reinterpret_cast<decltype(a.prepare_v2)>(sym("sqlite3_prepare_v2"));
        a.bind_int64 = reinterpret_cast<decltype(a.bind_int64)>(sym("sqlite3_bind_int64"));
        a.step = reinterpret_cast<decltype(a.step)>(sym("sqlite3_step"));
        a.column_double = reinterpret_cast<decltype(a.column_double)>(sym("sqlite3_column_double"));
        a.column_int64 = reinterpret_cast<decltype(a.column_int64)>(sym("sqlite3_column_int64"));
        a.column_text = reinterpret_cast<decltype(a.column_text)>(sym("sqlite3_column_text"));
        a.finalize = reinterpret_cast<decltype(a.finalize)>(sym("sqlite3_finalize"));
        a.ok = a.open_v2 && a.close && a.errmsg && a.prepare_v2 && a.bind_int64 && a.step && a.column_double &&
               a.column_int64 && a.column_text && a.finalize;
        return a;
    }();
    return api;
}

}   // namespace

SqliteDataLoader::SqliteDataLoader(const std::string &specFilePath, std::optional<size_t> maxLoadCount, bool verbose)
    : IDataLoader{maxLoadCount}, _verbose{verbose}
{
    loadColumnSpecData(specFilePath);
}

SqliteDataLoader::SqliteDataLoader(bool, const std::string &dbPath, std::optional<size_t> maxLoadCount)
    : IDataLoader{maxLoadCount}, _dbPath{dbPath}
{
}

SqliteDataLoader::~SqliteDataLoader()
{
    if (db && sqlite_api().ok)
        sqlite_api().close(db);
}

// column-spec file: one line per column, "<name>\t<weight>[\tbinary]" (SqliteDataLoader.cpp:11-67);
// selects the table "ican" like the reference
void SqliteDataLoader::loadColumnSpecData(const std::string &path)
{
    std::cerr << "Use of deprecated method SqliteDataLoader::loadColumnSpecData(const std::string &path)\n";
    std::ifstream in(path);
    std::string line;
    std::vector<ColumnSpec> spec;
    while (std::getline(in, line)) {
        std::istringstream fields(line);
        std::string token, name;
        float weight = 1.0f;
        bool isBinary = false;
        int column = 0;
        while (std::getline(fields, token, '\t')) {
            if (column == 0) {
                if (!token.empty() && std::isalnum((unsigned char)token[0]))
                    name = token;
            } else {
                char *end = nullptr;
                weight = std::strtof(token.c_str(), &end);
                if (*end) {                 // not a number: weight 1, and "binary" marks the column
                    weight = 1.0f;
                    if (token == "binary")
                        isBinary = true;
                }
            }
            ++column;
        }
        spec.emplace_back(name, weight, isBinary ? 1 : 0);
        if (_verbose)
            std::cout << name << "\tWeight:" << weight << "\tBinary:" << isBinary << "\n";
    }
    setColumnSpec(spec);
    _tableName = "ican";
}

void SqliteDataLoader::setColumnSpec(const std::vector<ColumnSpec> columnSpec) noexcept
{
    _columnNames.clear();
    _isBinary.clear();
    _isContinuous.clear();
    _columnSpec.clear();
    for (const auto &s : columnSpec) {     // SqliteDataLoader.cpp:74-101
        _columnNames.push_back(s.name);
        _isBinary.push_back(s.isBinary);
        _isContinuous.push_back(!s.isBinary);
        _columnSpec.emplace_back(s.name, s.weight, s.isBinary);
    }
    vectorLength = _columnNames.size();
}

const std::vector<float> SqliteDataLoader::getWeights() const noexcept
{
    std::vector<float> w;
    for (const auto &s : _columnSpec)
        w.push_back(s.weight);
    return w;
}

const std::vector<std::string> SqliteDataLoader::getNames() const noexcept
{
    std::vector<std::string> names;
    for (const auto &s : _columnSpec)
        names.push_back(s.name);
    return names;
}

bool SqliteDataLoader::open(const char *fileName)
{
    const auto &api = sqlite_api();
    if (!api.ok) {
        std::cout << "Cannot open database: libsqlite3.so.0 is not available\n";
        return false;
    }
    if (_verbose)
        std::cout << "Opening database: " << fileName << "\n";
    if (api.open_v2(fileName, &db, kSqliteOpenReadonly, nullptr) != kSqliteOk) {   // read-only, :134-154
        std::cout << "Cannot open database: " << (db ? api.errmsg(db) : "out of memory") << "\n";
        if (db)
            api.close(db);
        db = nullptr;
        return false;
    }
    hasOpenDatabase = true;
    return true;
}

std::vector<std::string> SqliteDataLoader::queryStrings(const std::string &sql)
{
    std::vector<std::string> out;
    const auto &api = sqlite_api();
    sqlite3_stmt *st = nullptr;
    if (api.prepare_v2(db, sql.c_str(), (int)sql.size(), &st, nullptr) != kSqliteOk) {
        std::cerr << "Failed to execute statement: " << sql << "\nError message:" << api.errmsg(db) << "\n";
        api.finalize(st);
        return out;
    }
    while (api.step(st) == kSqliteRow) {
        const unsigned char *t = api.column_text(st, 0);
        out.emplace_back(t ? reinterpret_cast<const char *>(t) : "NULL");
    }
    api.finalize(st);
    return out;
}

long long SqliteDataLoader::queryInteger(const std::string &sql)
{
    const auto r = queryStrings(sql);
    // the reference parses the text of the first column with strtoul ("NULL" -> 0), :373-417
    return r.empty() ? 0 : (long long)std::strtoul(r[0].c_str(), nullptr, 10);
}

std::vector<std::string> SqliteDataLoader::findAllColumns()
{
    if (!hasOpenDatabase) {
        std::cerr << "No open database - cannot parse columns\n";
        return {};
    }
    _columnNames = queryStrings("select name from pragma_table_info('" + _tableName + "') as tblInfo;");   // :276-293
    return _columnNames;
}

std::vector<std::string> SqliteDataLoader::findAllTables()
{
    if (!hasOpenDatabase) {
        std::cerr << "No open database - cannot parse columns\n";
        return {};
    }
    _tableNames = queryStrings("SELECT name FROM sqlite_master WHERE type = 'table' AND name NOT LIKE 'sqlite_%';");
    return _tableNames;
}

// SqliteDataLoader.cpp:481-548 (fetchData2): one range query per chunk
SqliteDataLoader::Fetch SqliteDataLoader::fetch(std::optional<long long> startId, std::optional<size_t> maxCount)
{
    if (_tableName.empty()) {
        std::cerr << "Cannot read database - No table name specified\n";
        return Fetch{{}, std::nullopt, 0};
    }
    if (!hasOpenDatabase) {
        std::cerr << "Cannot fetch data: no open database\n";
        return Fetch{{}, std::nullopt, 0};
    }
    const auto &api = sqlite_api();
    const size_t depth = getDepth();
    const long long maxIdValue = queryInteger("SELECT MAX(Id) FROM " + _tableName + ";");
    long long cur = startId.has_value() ? *startId : queryInteger("SELECT MIN(Id) FROM " + _tableName + ";");
    long long last = maxIdValue;
    if (maxCount.has_value())
        last = (cur + (long long)*maxCount) > maxIdValue ? maxIdValue : cur + (long long)*maxCount - 1;   // :492-493

    std::string sql = "SELECT ";
    for (const auto &c : getNames())
        sql += c + ",";
    sql += "Id FROM " + _tableName + " WHERE Id>=? AND Id <=?;";

    Fetch out{{}, std::nullopt, maxIdValue};
    sqlite3_stmt *st = nullptr;
    if (api.prepare_v2(db, sql.c_str(), (int)sql.size(), &st, nullptr) != kSqliteOk) {
        std::cerr << "Cannot fetch data: " << api.errmsg(db) << '\n';
        api.finalize(st);
        return Fetch{{}, std::nullopt, 0};
    }
    api.bind_int64(st, 1, cur);
    api.bind_int64(st, 2, last);
    while (api.step(st) == kSqliteRow) {
        RowData row;
        row.values = Eigen::VectorXf((Eigen::Index)depth);
        row.valid.assign(depth, 1);                                    // NULLs read as 0.0 and stay "valid" (:527-533)
        for (size_t c = 0; c < depth; ++c)
            row.values[(Eigen::Index)c] = (float)api.column_double(st, (int)c);
        cur = api.column_int64(st, (int)depth);
        out.rows.push_back(std::move(row));
    }
    api.finalize(st);
    std::cout << "Fetched whole batch\n";
    out.nextId = cur + 1;                                              // :547
    return out;
}

size_t SqliteDataLoader::load()
{
    Fetch f = fetch(currentLoadId, m_maxLoadCount);
    data = std::move(f.rows);
    currentLoadId = f.nextId;
    if (_verbose)
        std::cout << "\rLoading database:100%\n";
    if (currentLoadId.has_value() && *currentLoadId > f.maxId)         // back at the start of the stream (:475-476)
        currentLoadId.reset();
    return data.size();
}

std::vector<RowData> SqliteDataLoader::getPreview(size_t)
{
    return fetch(std::nullopt, 100).rows;                              // :618-622: always the first 100 ids
}
