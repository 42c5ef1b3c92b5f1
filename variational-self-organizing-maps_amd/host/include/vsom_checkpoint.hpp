// vsom_checkpoint.hpp -- the reference's Octave/Matlab text checkpoint (Som::save / Som::load /
// Som::getSizeFromFile, src/Som.cpp:1209-1597) as plain functions over host arrays, so that maps
// trained by the reference load into this build and vice versa.  No device
// code: Som::save/load (vsom_host.cpp) move the state between HBM and these arrays.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace vsom {

struct Checkpoint {
    size_t width = 0, height = 0, depth = 0;
    std::vector<float> map, sigma;       // row-major N x depth (node = y*width + x)
    std::vector<float> weight;           // N
    std::vector<uint64_t> hits;          // N
    std::vector<double> U;               // N
    void resize()
    {
        const size_t N = width * height;
        map.assign(N * depth, 0.f);
        sigma.assign(N * depth, 0.f);
        weight.assign(N, 0.f);
        hits.assign(N, 0);
        U.assign(N, 0.0);
    }
};

// Som::save (Som.cpp:1209-1294): same bytes (printf "%f" / "%lu" formatting, section order,
// element-major "som"/"sigmaSom" blocks).  Returns false when the file cannot be opened.
bool write_octave(const char *fileName, const Checkpoint &c);

// Som::getSizeFromFile (Som.cpp:1296-1341): depth from the line after "# ndims: ", height from
// "# rows: ", width from "# columns: " (the last occurrence wins).
bool read_octave_dims(const char *fileName, size_t &width, size_t &height, size_t &depth);

// Som::load (Som.cpp:1343-1597) into a Checkpoint whose arrays are already sized (c.resize()); on
// the way it reassigns width = "# rows", height = "# columns" like the reference.  Throws
// std::runtime_error where the reference calls exit (unknown section name, wrong type).
bool read_octave(const char *fileName, Checkpoint &c);

}   // namespace vsom
