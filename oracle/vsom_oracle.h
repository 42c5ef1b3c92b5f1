/*
 * vsom_oracle.h -- CPU ORACLE for the VSOM training hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the reference algorithm
 * (PereUbu7/Variational-Self-Organizing-Maps: src/Som.cpp, src/Transformation.cpp,
 * src/SomIndex.cpp, src/DataSet.cpp).  It is the checker the HIP path is compared
 * against.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * link or call it.  The product (libvsom_hip.so and the host API above it) never does.
 *
 * PARITY UNPINNED: the reference cannot be compiled in this image (Eigen 3, doctest
 * and extern/sqlite/sqlite3.c are absent; no network) and its own tests assert no
 * numeric result for this path (tests/test1.cpp prints only).  The oracle is therefore
 * pinned only by (a) hand-derived known answers from the reference source
 * (tests/golden/kat.json: CLR pair order of tests/test1.cpp:18-43, neighbourhood
 * weights, accumulator recurrences) and (b) its own committed golden vectors.  The
 * fp32 summation order of Eigen's `dot` (Q1 below) is restated from Eigen 3.4's
 * linear-vectorised reduction with SSE packets of 4 floats; the Eigen version is not
 * pinned by the reference, so this order is a documented convention.
 *
 * All model arithmetic is fp32 with one rounding per operation (no FMA contraction:
 * the reference is built -msse2 without -mfma, build/Makefile:5,18); neighbourhood
 * weights, eta and sigma are double.  Compile with -ffp-contract=off.
 */
#ifndef VSOM_ORACLE_H
#define VSOM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Transformation kinds: src/Transformation.cpp:3-39, 41-77, 79-167 */
enum { VSO_STANDARD = 0, VSO_MEDIAN = 1, VSO_CLR = 2 };
/* Som::WeigthDecayFunction: include/SOM.hpp:70-75 */
enum { VSO_EXPONENTIAL = 0, VSO_INVERSE_PROPORTIONAL = 1, VSO_BATCHMAP = 2 };

/* State of `class Som` (include/SOM.hpp:55-64) in contiguous row-major arrays. */
typedef struct vso_som {
    size_t width, height;
    size_t depth;    /* D = Length(J): model-vector length          */
    size_t in_len;   /* J: input-vector length                      */
    int transform;
    float *map;      /* N x D   Som::map                            */
    float *sigma;    /* N x D   Som::sigmaMap                       */
    float *S;        /* N x D   Som::SMap                           */
    float *weight;   /* N       Som::weightMap                      */
    uint64_t *hits;  /* N       Som::bmuHits                        */
} vso_som;

/* Transformation::Length  (Transformation.cpp:31-35, 69-73, 162-165) */
size_t vso_length(int transform, size_t in_len);

/* Som::Construct (Som.cpp:11-48): everything zero. */
vso_som *vso_create(size_t width, size_t height, size_t in_len, int transform);
void vso_free(vso_som *som);

/* Som::randomInitialize (Som.cpp:977-997): glibc srand/rand, node-major, dim-minor. */
void vso_random_initialize(vso_som *som, int seed, float sigma);

/* Transformation::Comparer -> out[vso_comparer_len] ; Transformation::Stepper -> out[D] */
size_t vso_comparer_len(int transform, size_t depth);
void vso_comparer(int transform, const float *value, size_t in_len,
                  const float *model, size_t depth, float *out);
void vso_stepper(int transform, const float *value, size_t in_len,
                 const float *model, size_t depth, float *out);

/* r.dot(r) in Eigen's SSE linear-vectorised reduction order. */
float vso_dot_self(const float *r, size_t n);

/* SomIndex(const Som&, size_t) (SomIndex.cpp:13-18): y divides by HEIGHT. */
void vso_somindex(const vso_som *som, size_t index, size_t *x, size_t *y);

/* Som::euclidianWeightedDist (Som.cpp:124-141). */
double vso_dist(const vso_som *som, size_t node, const float *v);
/* Som::findBmu (Som.cpp:291-309) -> linear index y*W+x. */
size_t vso_find_bmu(const vso_som *som, const float *v);
/* Som::findLocalBmu (Som.cpp:335-454) -> linear index. */
size_t vso_find_local_bmu(const vso_som *som, const float *v, size_t last_bmu);
/* Som::calculateNeighbourhoodWeight (Som.cpp:949-975). */
double vso_neighbourhood_weight(size_t cx, size_t cy, size_t bx, size_t by, double sigma);

/*
 * Som::trainBatchSomEpoch (Som.cpp:756-879), split so that shards can be checked:
 *  phase1_range : samples [s0,s1): lastbmu[s], sqres[s] = ||Comparer(x_s, M[bmu])||^2
 *  phase1_finish: bmuHits += and the fp32 MSE running sum, in sample order
 *  phase2_range : nodes [n0,n1): new map / sigmaMap / weightMap rows
 * X is B x in_len row-major.  nthreads<=1 -> serial.
 */
void vso_batch_phase1_range(const vso_som *som, const float *X, size_t B,
                            size_t s0, size_t s1, uint64_t *lastbmu, float *sqres,
                            int is_first, int nthreads);
float vso_batch_phase1_finish(vso_som *som, const uint64_t *lastbmu,
                              const float *sqres, size_t B);
void vso_batch_phase2_range(vso_som *som, const float *X, size_t B,
                            const uint64_t *lastbmu, double sigma,
                            size_t n0, size_t n1, int nthreads);
float vso_batch_epoch(vso_som *som, const float *X, size_t B, uint64_t *lastbmu,
                      double sigma, int is_first, int nthreads);

/*
 * Som::trainBatchSom (Som.cpp:716-754) over `nchunks` chunks (chunk c = rows
 * [chunk_off[c], chunk_off[c+1]) of X).  lastBMU is zeroed at every chunk load
 * (DataSet.cpp:136-137).  mse_out[epochs] (entries after an early sigma<1 return
 * are left untouched).  Returns the number of epochs actually run.
 */
size_t vso_train_batch(vso_som *som, const float *X, const size_t *chunk_off,
                       size_t nchunks, size_t epochs, double sigma0,
                       double sigma_decay, float *mse_out, int nthreads);

/*
 * Som::trainSingle (Som.cpp:885-947).  residual_out has vso_comparer_len entries
 * (may be NULL).  Returns the BMU linear index; *last_bmu is updated.
 */
size_t vso_train_single(vso_som *som, const float *v, double eta, double sigma,
                        uint64_t *last_bmu, int decay_fn,
                        float *residual_out, float *dist_out);
/* Inner loop of Som::trainBasicSom over one chunk (Som.cpp:1159-1171): returns the
 * chunk's fp32 MSE contribution; lastbmu[B] in/out; bmuHits updated (addBmu). */
/* running-sum form: returns mse_start + the chunk's terms, added in sample order (Som.cpp:1153,1167) */
float vso_train_online_chunk_from(vso_som *som, const float *X, size_t B, uint64_t *lastbmu,
                                  double eta, double sigma, int decay_fn, float mse_start);
float vso_train_online_chunk(vso_som *som, const float *X, size_t B,
                             uint64_t *lastbmu, double eta, double sigma, int decay_fn);
/* Som::trainBasicSom (Som.cpp:1135-1187). */
void vso_train_online(vso_som *som, const float *X, const size_t *chunk_off,
                      size_t nchunks, size_t epochs, double eta0, double eta_decay,
                      double sigma0, double sigma_decay, int decay_fn, float *mse_out);

/* ---- "next" rows of SURVEY 8f (consumers of the BMU search; U-matrix) ------------------------ */
/* Som::findRestrictedBmu (Som.cpp:313-332): node 0 seeds unconditionally. */
size_t vso_find_restricted_bmu(const vso_som *som, const float *v, uint64_t min_hits);
/* Som::findRestrictedBmd (Som.cpp:457-487): out[N] doubles. */
void vso_find_restricted_bmd(const vso_som *som, const float *v, uint64_t min_hits, double *out);
/* Som::euclidianWeightedDistRaw (Som.cpp:143-157) with valid == weights == 1. */
double vso_dist_raw(const vso_som *som, size_t pos, const float *v);
/* Som::updateUMatrix (Som.cpp:999-1111): U[N] doubles. */
void vso_update_umatrix(const vso_som *som, double *U);

/* CPU-baseline variant that mirrors the reference's per-call temporaries
 * (heap vector per Comparer/Stepper call, neuron copy) -- timing honesty only,
 * results identical to vso_batch_epoch(nthreads=1). */
float vso_batch_epoch_faithful(vso_som *som, const float *X, size_t B,
                               uint64_t *lastbmu, double sigma, int is_first);

int vso_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
