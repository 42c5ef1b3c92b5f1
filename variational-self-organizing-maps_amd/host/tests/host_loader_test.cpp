// host_loader_test -- drives MnistDataLoader / SqliteDataLoader through DataSet the way the
// reference's drivers do (loadNextDataFromStream until hasReadWholeDataStream) and prints every
// chunk; tests/test_host_loaders.py builds the input files and checks the output.  No GPU needed.
//   host_loader_test mnist  <folder> <maxLoadCount|0>
//   host_loader_test sqlite <db> <table> <maxLoadCount|0> <col,col,...>
//   host_loader_test spec   <db> <specfile> <maxLoadCount|0>
//   host_loader_test octave <in.txt> <out.txt>      (reference checkpoint text: read dims, load, save)
#include "DataSet.hpp"
#include "MnistDataLoader.hpp"
#include "SqliteDataLoader.hpp"
#include "vsom_checkpoint.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>

static void dump_stream(IDataLoader &loader, int passes)
{
    DataSet ds(loader);
    for (int pass = 0; pass < passes; ++pass) {
        while (!ds.hasReadWholeDataStream()) {
            ds.loadNextDataFromStream();
            std::printf("CHUNK %zu %zu\n", ds.size(), ds.vectorLength());
            const float *flat = ds.contiguous();
            for (size_t r = 0; r < ds.size(); ++r) {
                std::printf("ROW");
                for (size_t d = 0; d < ds.vectorLength(); ++d)
                    std::printf(" %.9g", (double)flat[r * ds.vectorLength() + d]);
                std::printf(" | lastBMU %zu valid %d\n", ds.getLastBMU(r), ds.getValidity(r)[0]);
            }
        }
        ds.resetStreamLoadPosition();
        std::printf("PASS_END\n");
    }
}

int main(int argc, char **argv)
{
    if (argc < 4)
        return 2;
    const std::string mode = argv[1];
    if (mode == "octave") {
        vsom::Checkpoint c;
        if (!vsom::read_octave_dims(argv[2], c.width, c.height, c.depth))
            return 3;
        std::printf("DIMS %zu %zu %zu\n", c.width, c.height, c.depth);
        c.resize();
        if (!vsom::read_octave(argv[2], c))
            return 3;
        return vsom::write_octave(argv[3], c) ? 0 : 4;
    }
    auto count = [](const char *s) -> std::optional<size_t> {
        const long v = std::atol(s);
        return v > 0 ? std::optional<size_t>((size_t)v) : std::nullopt;
    };
    if (mode == "mnist") {
        MnistDataLoader loader(count(argv[3]));
        loader.open(argv[2]);
        std::printf("DEPTH %zu NAME0 %s NAME783 %s NAME784 %s\n", loader.getDepth(), loader.getName(0).c_str(),
                    loader.getName(783).c_str(), loader.getName(784).c_str());
        dump_stream(loader, 2);
        return 0;
    }
    if (mode == "sqlite" && argc >= 6) {
        SqliteDataLoader loader(true, argv[2], count(argv[4]));
        if (!loader.open())
            return 3;
        for (const auto &t : loader.findAllTables())
            std::printf("TABLE %s\n", t.c_str());
        loader.setTable(argv[3]);
        for (const auto &c : loader.findAllColumns())
            std::printf("COLUMN %s\n", c.c_str());
        std::vector<ColumnSpec> spec;
        std::stringstream ss(argv[5]);
        std::string col;
        while (std::getline(ss, col, ','))
            spec.emplace_back(col, 1.0f, 0);
        loader.setColumnSpec(spec);
        dump_stream(loader, 2);
        return 0;
    }
    if (mode == "spec" && argc >= 5) {
        SqliteDataLoader loader(argv[3], count(argv[4]));
        if (!loader.open(argv[2]))
            return 3;
        for (size_t i = 0; i < loader.getDepth(); ++i)
            std::printf("SPEC %s %g %d\n", loader.getName(i).c_str(), (double)loader.getWeight(i), loader.getBinary()[i]);
        dump_stream(loader, 1);
        return 0;
    }
    return 2;
}
