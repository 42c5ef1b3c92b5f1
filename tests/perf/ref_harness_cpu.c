/*
 * ref_harness_cpu.c -- the reference's own performance-harness scenarios (tests/performance/perf_tests.cpp:74-402)
 * on the CPU ORACLE (oracle/vsom_oracle.h), one thread, as the figure printed beside the device path's
 * (variational-self-organizing-maps_amd/host/tests/host_api_test.cpp, mode ref_harness).  TEST INFRASTRUCTURE: it links
 * the oracle, so it lives under tests/ and nothing of the product builds or loads it.
 *
 * The oracle restates the arithmetic of the hot path, not the reference's containers: its per-call cost is a LOWER bound
 * of what libsom spends (no Eigen temporaries, no std::function calls); the scenarios that wrap a search in host-side
 * post-processing (evaluate, measureSimilarity, variationalAutoEncoder) are timed as the searches they make.
 *   usage: ref_harness_cpu <fixture_rows.f32> [scale]
 */
#define _POSIX_C_SOURCE 199309L
#include "../../oracle/vsom_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_us(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}

static void line(const char *scenario, size_t calls, double us_total, const char *unit)
{
    printf("{\"scenario\": \"%s\", \"calls\": %zu, \"cpu_oracle_%s\": %.3f}\n", scenario, calls, unit, us_total / (double)calls);
    fflush(stdout);
}

static float frand(void) { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; }   /* Eigen::VectorXf::Random: [-1, 1] */

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: ref_harness_cpu <fixture_rows.f32> [scale]\n");
        return 2;
    }
    const size_t scale = argc > 2 ? (size_t)strtoul(argv[2], NULL, 10) : 100;
    enum { FR = 20, FJ = 9 };
    float fx[FR * FJ];
    FILE *f = fopen(argv[1], "rb");
    if (!f || fread(fx, 4, FR * FJ, f) != FR * FJ) {
        fprintf(stderr, "cannot read %s\n", argv[1]);
        return 2;
    }
    fclose(f);
    const size_t runs = 100;
    volatile double sink = 0;
    double t0;

    {   /* Som::randomInitialize (:162-178) */
        vso_som *s = vso_create(100, 100, 100, VSO_STANDARD);
        double *U = (double *)malloc(10000 * sizeof(double));
        vso_random_initialize(s, 1, 1.f);
        t0 = now_us();
        vso_random_initialize(s, 1, 1.f);
        vso_update_umatrix(s, U);
        for (size_t i = 0; i < runs; ++i)
            vso_random_initialize(s, 1, 1.f);
        line("Som::randomInitialize() 100x100x100", runs, now_us() - t0, "us_per_call");
        free(U);
        vso_free(s);
    }
    {   /* Som::train x 3 (:74-112) */
        const size_t off[2] = {0, FR};
        float mse[300];
        const int fns[3] = {VSO_EXPONENTIAL, VSO_BATCHMAP, VSO_INVERSE_PROPORTIONAL};
        const char *names[3] = {"Som::train(exponentialWeightDecay) 10x10x9, 20 rows, 300 epochs",
                                "Som::train(batchMap) 10x10x9, 20 rows, 300 epochs asked (sigma < 1 ends it after 231)",
                                "Som::train(inverseProportionalWeightDecay) 10x10x9, 20 rows, 300 epochs"};
        for (int m = 0; m < 3; ++m) {
            vso_som *s = vso_create(10, 10, FJ, VSO_STANDARD);
            vso_random_initialize(s, 7, 1.f);
            t0 = now_us();
            if (fns[m] == VSO_BATCHMAP)
                (void)vso_train_batch(s, fx, off, 1, 300, 10.0, 0.01, mse, 1);
            else
                vso_train_online(s, fx, off, 1, 300, 0.001, 0.01, 10.0, 0.01, fns[m], mse);
            line(names[m], 300, now_us() - t0, "us_per_epoch");
            vso_free(s);
        }
        {   /* the batch epochs again in the oracle's `faithful` mode: the reference's per-call temporaries (a heap vector per
             * Comparer / Stepper call, a neuron copy per distance) -- closer to what libsom itself spends per epoch */
            vso_som *s = vso_create(10, 10, FJ, VSO_STANDARD);
            uint64_t lbf[FR];
            vso_random_initialize(s, 7, 1.f);
            size_t done = 0;
            t0 = now_us();
            for (size_t e = 0; e < 300; ++e) {
                const double sigma = 10.0 * exp(-0.01 * (double)e);
                if (sigma < 1.0)
                    break;
                memset(lbf, 0, sizeof(lbf));
                sink += vso_batch_epoch_faithful(s, fx, FR, lbf, sigma, e == 0);
                ++done;
            }
            printf("{\"scenario\": \"%s\", \"calls\": %zu, \"cpu_oracle_faithful_us_per_epoch\": %.3f}\n", names[1], done,
                   (now_us() - t0) / (double)done);
            fflush(stdout);
            vso_free(s);
        }
    }
    {   /* evaluate / measureSimilarity / variationalAutoEncoder: the searches they make, 100x100x9 over 20 rows */
        vso_som *s = vso_create(100, 100, FJ, VSO_STANDARD);
        double *bmd = (double *)malloc(10000 * sizeof(double));
        vso_random_initialize(s, 7, 1.f);
        t0 = now_us();
        for (size_t r = 0; r < runs; ++r)
            for (size_t i = 0; i < FR; ++i) {
                const size_t b = vso_find_bmu(s, fx + i * FJ);
                sink += vso_dist(s, b, fx + i * FJ);
            }
        line("Som::evaluate() 100x100x9 over 20 rows", runs, now_us() - t0, "us_per_call");
        t0 = now_us();
        for (size_t r = 0; r < runs; ++r)
            for (size_t i = 0; i < FR; ++i)
                sink += (double)vso_find_restricted_bmu(s, fx + i * FJ, 1);
        line("Som::measureSimilarity() 100x100x9 over 20 rows", runs, now_us() - t0, "us_per_call");
        t0 = now_us();
        for (size_t r = 0; r < runs; ++r)
            for (size_t i = 0; i < FR; ++i) {
                vso_find_restricted_bmd(s, fx + i * FJ, 0, bmd);
                sink += bmd[17];
            }
        line("Som::variationalAutoEncoder() 100x100x9 over 20 rows", runs, now_us() - t0, "us_per_call");
        free(bmd);
        vso_free(s);
    }
    {   /* updateUMatrix, CLR map of depth 72 (:320-344) */
        vso_som *s = vso_create(100, 100, FJ, VSO_CLR);
        double *U = (double *)malloc(10000 * sizeof(double));
        vso_random_initialize(s, 7, 1.f);
        t0 = now_us();
        for (size_t r = 0; r < runs; ++r)
            vso_update_umatrix(s, U);
        sink += U[5];
        line("Som::updateUMatrix() 100x100 CLR map, depth 72", runs, now_us() - t0, "us_per_call");
        free(U);
        vso_free(s);
    }
    {   /* single-vector calls, 100x100x100 (:114-140, 181-295) */
        vso_som *s = vso_create(100, 100, 100, VSO_STANDARD);
        double *bmd = (double *)malloc(10000 * sizeof(double));
        float *samples = (float *)malloc(1000 * 100 * sizeof(float)), mv[100];
        uint64_t positions[1000];
        srand(12345);
        for (size_t i = 0; i < 1000 * 100; ++i)
            samples[i] = frand();
        for (size_t i = 0; i < 100; ++i)
            mv[i] = frand();
        for (size_t i = 0; i < 1000; ++i)
            positions[i] = (uint64_t)(rand() % 100 * 100);
        for (int m = 0; m < 2; ++m) {
            vso_random_initialize(s, 7, 1.f);
            uint64_t pos[1000];
            memcpy(pos, positions, sizeof(pos));
            t0 = now_us();
            for (size_t i = 0; i < 1000; ++i)
                sink += (double)vso_train_single(s, samples + i * 100, 0.1, 50.0, &pos[i],
                                                 m == 0 ? VSO_EXPONENTIAL : VSO_INVERSE_PROPORTIONAL, NULL, NULL);
            line(m == 0 ? "Som::trainSingle(exponentialWeightDecay) 100x100x100, sigma 50"
                        : "Som::trainSingle(inverseProportionalWeightDecay) 100x100x100, sigma 50",
                 1000, now_us() - t0, "us_per_call");
        }
        vso_random_initialize(s, 7, 1.f);
        {
            const size_t n = 1000000 / scale;
            t0 = now_us();
            for (size_t i = 0; i < n; ++i)
                sink += vso_dist(s, (size_t)positions[i % 1000], mv) / 1000000;
            line("Som::euclidianWeightedDist() 100x100x100", n, now_us() - t0, "us_per_call");
        }
        t0 = now_us();
        for (size_t i = 0; i < 1000; ++i)
            sink += (double)vso_find_bmu(s, mv);
        line("Som::findBmu() 100x100x100", 1000, now_us() - t0, "us_per_call");
        {
            const size_t n = 1000000 / scale;
            t0 = now_us();
            for (size_t i = 0; i < n; ++i)
                sink += (double)vso_find_local_bmu(s, mv, (size_t)positions[i % 1000]);
            line("Som::findLocalBmu() 100x100x100", n, now_us() - t0, "us_per_call");
        }
        t0 = now_us();
        for (size_t i = 0; i < 1000; ++i)
            sink += (double)vso_find_restricted_bmu(s, mv, 1);
        line("Som::findRestrictedBmu(minBmuHits 1) 100x100x100", 1000, now_us() - t0, "us_per_call");
        t0 = now_us();
        for (size_t i = 0; i < 1000; ++i) {
            vso_find_restricted_bmd(s, mv, 0, bmd);
            sink += bmd[3];
        }
        line("Som::findRestrictedBmd(minBmuHits 0) 100x100x100", 1000, now_us() - t0, "us_per_call");
        free(bmd);
        free(samples);
        vso_free(s);
    }
    fprintf(stderr, "%g\n", (double)sink);
    return 0;
}
