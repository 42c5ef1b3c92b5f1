/*
 * vsom_oracle.c -- CPU ORACLE (test infrastructure, see vsom_oracle.h header).
 * PARITY UNPINNED (reference unbuildable here; no numeric assertions in its tests).
 *
 * Every function cites the reference lines it restates (paths relative to the
 * reference repository root).  fp32 arithmetic, one rounding per operation; build
 * with -O2 -msse2 -ffp-contract=off (never -ffast-math).
 */
#include "vsom_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define VSO_MIN(a, b) ((a) < (b) ? (a) : (b))
#define VSO_MAX(a, b) ((a) > (b) ? (a) : (b))

int vso_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ---- Transformation::Length: Transformation.cpp:31-35,69-73,162-165 ---- */
size_t vso_length(int transform, size_t in_len)
{
    if (transform == VSO_CLR)
        return in_len * (in_len - 1u);
    return in_len;
}

size_t vso_comparer_len(int transform, size_t depth)
{
    /* CLR Comparer returns A.size() = model.size()/2 entries (Transformation.cpp:87,104) */
    return transform == VSO_CLR ? depth / 2 : depth;
}

/* ---- Som::Construct: Som.cpp:11-48 ---- */
vso_som *vso_create(size_t width, size_t height, size_t in_len, int transform)
{
    vso_som *s = (vso_som *)calloc(1, sizeof(vso_som));
    if (!s)
        return NULL;
    s->width = width;
    s->height = height;
    s->in_len = in_len;
    s->depth = vso_length(transform, in_len);
    s->transform = transform;
    size_t n = width * height, nd = n * s->depth;
    s->map = (float *)calloc(nd ? nd : 1, sizeof(float));
    s->sigma = (float *)calloc(nd ? nd : 1, sizeof(float));
    s->S = (float *)calloc(nd ? nd : 1, sizeof(float));
    s->weight = (float *)calloc(n ? n : 1, sizeof(float));
    s->hits = (uint64_t *)calloc(n ? n : 1, sizeof(uint64_t));
    if (!s->map || !s->sigma || !s->S || !s->weight || !s->hits) {
        vso_free(s);
        return NULL;
    }
    return s;
}

void vso_free(vso_som *s)
{
    if (!s)
        return;
    free(s->map);
    free(s->sigma);
    free(s->S);
    free(s->weight);
    free(s->hits);
    free(s);
}

/* ---- Som::randomInitialize: Som.cpp:977-997 ---- */
void vso_random_initialize(vso_som *som, int seed, float sigma)
{
    srand((unsigned)seed);
    size_t n = som->width * som->height, D = som->depth;
    for (size_t i = 0; i < n; ++i) {
        for (size_t d = 0; d < D; ++d) {
            /* (float)(rand() % (int)(2000*sigma)) - 1000.f*sigma) / 1000.f : Som.cpp:988 */
            som->map[i * D + d] =
                ((float)(rand() % (int)(2000 * sigma)) - (1000.f * sigma)) / 1000.f;
            som->sigma[i * D + d] = 0.0f;
            som->S[i * D + d] = 0.0f;
        }
        som->weight[i] = 0.0f;
        som->hits[i] = 0u;
    }
}

/* ---- Transformation::Comparer ----
 * Standard / Median: model - value            (Transformation.cpp:7-8, 45-46)
 * CLR: A.*x' + B - y', pairs i<j lexicographic (Transformation.cpp:82-106)
 * dispersion and valueWeight are ignored by every built-in.
 */
void vso_comparer(int transform, const float *value, size_t in_len,
                  const float *model, size_t depth, float *out)
{
    if (transform == VSO_CLR) {
        size_t P = depth / 2;
        const float *A = model, *Bm = model + P; /* head / tail: :87-88 */
        size_t p = 0;
        for (size_t i = 0; i < in_len; ++i) {
            for (size_t j = i + 1; j < in_len; ++j) { /* :95-101 */
                if (p < P) {
                    float t = A[p] * value[i]; /* A.array()*xPrime.array()     :104 */
                    t = t + Bm[p];             /*  + B.array()                      */
                    t = t - value[j];          /*  - yPrime.array()                 */
                    out[p] = t;
                }
                ++p;
            }
        }
        return;
    }
    for (size_t d = 0; d < depth; ++d)
        out[d] = model[d] - value[d];
}

/* sign(): Eigen 3.4 scalar_sign_op<real>: NaN -> NaN, else (a>0)-(a<0). */
static inline float vso_signf(float a)
{
    if (a != a)
        return a;
    return (float)((a > 0.0f) - (a < 0.0f));
}

/* ---- Transformation::Stepper ----
 * Standard: value - model                                   (Transformation.cpp:11-12)
 * Median:   sign(value - model)                             (Transformation.cpp:49-50)
 * CLR:      [ (-2*inner).*x' ; -2*inner ], inner = A.*x'+B-y' (Transformation.cpp:107-142)
 */
void vso_stepper(int transform, const float *value, size_t in_len,
                 const float *model, size_t depth, float *out)
{
    if (transform == VSO_CLR) {
        size_t P = depth / 2;
        const float *A = model, *Bm = model + P;
        size_t p = 0;
        for (size_t i = 0; i < in_len; ++i) {
            for (size_t j = i + 1; j < in_len; ++j) {
                if (p < P) {
                    float inner = A[p] * value[i]; /* :129 */
                    inner = inner + Bm[p];
                    inner = inner - value[j];
                    float m2 = -2.0f * inner;  /* -2*inner            :135-136 (exact) */
                    out[p] = m2 * value[i];    /* (-2*inner)*xPrime   :135             */
                    out[P + p] = m2;           /* bDelta              :136, :140       */
                }
                ++p;
            }
        }
        return;
    }
    if (transform == VSO_MEDIAN) {
        for (size_t d = 0; d < depth; ++d)
            out[d] = vso_signf(value[d] - model[d]);
        return;
    }
    for (size_t d = 0; d < depth; ++d)
        out[d] = value[d] - model[d];
}

/*
 * ---- r.dot(r)  (Som.cpp:140) in Eigen 3.4's order ----
 * Eigen/src/Core/Redux.h, redux_impl<.., LinearVectorizedTraversal, NoUnrolling>
 * with SSE Packet4f (no FMA): two packet accumulators over blocks of 8, then
 * p0+=p1, one more whole packet if present, predux = (a0+a2)+(a1+a3)
 * (arch/SSE/PacketMath.h predux<Packet4f>: movehl + add_ss), then scalar tail.
 * Sizes < 4 are summed sequentially.  Products are rounded before adding.
 */
float vso_dot_self(const float *r, size_t n)
{
    if (n == 0)
        return 0.0f;
    size_t aligned2 = (n / 8) * 8, aligned = (n / 4) * 4;
    float res;
    if (aligned) {
        float p0[4], p1[4];
        for (int k = 0; k < 4; ++k)
            p0[k] = r[k] * r[k];
        if (aligned > 4) {
            for (int k = 0; k < 4; ++k)
                p1[k] = r[4 + k] * r[4 + k];
            for (size_t idx = 8; idx < aligned2; idx += 8) {
                for (int k = 0; k < 4; ++k) {
                    float a = r[idx + k] * r[idx + k];
                    p0[k] = p0[k] + a;
                }
                for (int k = 0; k < 4; ++k) {
                    float b = r[idx + 4 + k] * r[idx + 4 + k];
                    p1[k] = p1[k] + b;
                }
            }
            for (int k = 0; k < 4; ++k)
                p0[k] = p0[k] + p1[k];
            if (aligned > aligned2) {
                for (int k = 0; k < 4; ++k) {
                    float a = r[aligned2 + k] * r[aligned2 + k];
                    p0[k] = p0[k] + a;
                }
            }
        }
        float t02 = p0[0] + p0[2];
        float t13 = p0[1] + p0[3];
        res = t02 + t13;
        for (size_t idx = aligned; idx < n; ++idx) {
            float a = r[idx] * r[idx];
            res = res + a;
        }
    } else {
        res = r[0] * r[0];
        for (size_t idx = 1; idx < n; ++idx) {
            float a = r[idx] * r[idx];
            res = res + a;
        }
    }
    return res;
}

/* ---- SomIndex(const Som&, size_t): SomIndex.cpp:13-18 (divides by HEIGHT, Q10) ---- */
void vso_somindex(const vso_som *som, size_t index, size_t *x, size_t *y)
{
    *x = index % som->width;
    *y = (index - index % som->width) / som->height;
}

/* distance with caller scratch (comparer_len floats) */
static float dist_scratch(const vso_som *som, size_t node, const float *v, float *scratch)
{
    size_t D = som->depth;
    vso_comparer(som->transform, v, som->in_len, som->map + node * D, D, scratch);
    return vso_dot_self(scratch, vso_comparer_len(som->transform, D));
}

/* ---- Som::euclidianWeightedDist: Som.cpp:124-141 ----
 * sM (:131) and validEigen (:134) are computed by the reference but ignored by all
 * built-in Comparers, so they do not appear here (Q2).  Returns fp32 widened. */
double vso_dist(const vso_som *som, size_t node, const float *v)
{
    size_t L = vso_comparer_len(som->transform, som->depth);
    float *scratch = (float *)malloc((L ? L : 1) * sizeof(float));
    float d = dist_scratch(som, node, v, scratch);
    free(scratch);
    return (double)d;
}

static size_t find_bmu_scratch(const vso_som *som, const float *v, float *scratch)
{
    /* Som.cpp:293-304: init with node 0, scan 0..N-1, strict <  (Q3) */
    size_t N = som->width * som->height;
    double minDist = (double)dist_scratch(som, 0, v, scratch);
    size_t minIndex = 0;
    for (size_t i = 0; i < N; ++i) {
        double cur = (double)dist_scratch(som, i, v, scratch);
        if (cur < minDist) {
            minDist = cur;
            minIndex = i;
        }
    }
    return minIndex; /* SomIndex(min % W, min / W) -> y*W+x == minIndex (:306) */
}

/* ---- Som::findBmu: Som.cpp:291-309 ---- */
size_t vso_find_bmu(const vso_som *som, const float *v)
{
    size_t L = vso_comparer_len(som->transform, som->depth);
    float *scratch = (float *)malloc((L ? L : 1) * sizeof(float));
    size_t r = find_bmu_scratch(som, v, scratch);
    free(scratch);
    return r;
}

/* ---- Som::findLocalBmu: Som.cpp:335-454, unsigned arithmetic kept literal (Q5) ---- */
static size_t find_local_bmu_scratch(const vso_som *som, const float *v,
                                     size_t lastBMUref, float *scratch)
{
    const size_t width = som->width, height = som->height;
    size_t lastBMU = lastBMUref;
    double minDist = (double)dist_scratch(som, lastBMU, v, scratch);
    size_t minIndex = lastBMU;
    const size_t m1 = (size_t)-1; /* -1uz */
    const size_t firstSearchX[8] = {m1, 0, 1, 1, 1, 0, m1, m1}; /* :341 */
    const size_t firstSearchY[8] = {1, 1, 1, 0, m1, m1, m1, 0}; /* :342 */

    size_t lastMeasured = lastBMU;
    size_t lastMeasuredX, lastMeasuredY, lastBMUX, lastBMUY;
    size_t currentX, currentY;
    size_t startX, endX;

    for (;;) {
        lastMeasuredX = lastMeasured % width;
        lastMeasuredY = lastMeasured / width;
        lastBMUX = lastBMU % width;
        lastBMUY = lastBMU / width;

        if (lastMeasured == lastBMU) { /* first try :362-385 */
            for (size_t i = 0; i < 8; ++i) {
                /* std::max(std::min(a + off, W-1), 0uz): max is a no-op on size_t */
                currentX = VSO_MAX(VSO_MIN(lastMeasuredX + firstSearchX[i], width - 1), (size_t)0);
                currentY = VSO_MAX(VSO_MIN(lastMeasuredY + firstSearchY[i], height - 1), (size_t)0);
                double cur = (double)dist_scratch(som, currentY * width + currentX, v, scratch);
                if (cur < minDist) {
                    minDist = cur;
                    minIndex = currentY * width + currentX;
                }
            }
            if (minIndex == lastBMU)
                return minIndex;
            lastMeasured = minIndex;
        } else { /* :387-450 */
            if (lastMeasuredX - lastBMUX) { /* moving in X :390-403 */
                for (int i = -1; i < 2; ++i) {
                    currentX = VSO_MAX(VSO_MIN(lastMeasuredX + lastMeasuredX - lastBMUX, width - 1), (size_t)0);
                    currentY = VSO_MAX(VSO_MIN(lastMeasuredY + (size_t)i, height - 1), (size_t)0);
                    double cur = (double)dist_scratch(som, currentY * width + currentX, v, scratch);
                    if (cur < minDist) {
                        minDist = cur;
                        minIndex = currentY * width + currentX;
                    }
                }
            }
            if (lastMeasuredY - lastBMUY) { /* moving in Y :406-437 */
                if (lastMeasuredX - lastBMUX > 0) { /* unsigned: "!= 0" */
                    startX = m1;
                    endX = 0;
                } else if (0) { /* `lastMeasuredX - lastBMUX < 0` on size_t: never true (:415) */
                    startX = 0;
                    endX = 1;
                } else {
                    startX = m1;
                    endX = 1;
                }
                /* starts at SIZE_MAX: the body never executes (:426) */
                for (size_t i = startX; i < (endX + 1); ++i) {
                    currentX = VSO_MAX(VSO_MIN(lastMeasuredX + i, width - 1), (size_t)0);
                    currentY = VSO_MAX(VSO_MIN(lastMeasuredY + lastMeasuredY - lastBMUY, height - 1), (size_t)0);
                    double cur = (double)dist_scratch(som, currentY * width + currentX, v, scratch);
                    if (cur < minDist) {
                        minDist = cur;
                        minIndex = currentY * width + currentX;
                    }
                }
            }
            if (minIndex == lastMeasured)
                return minIndex;
            lastBMU = lastMeasured;
            lastMeasured = minIndex;
        }
    }
}

size_t vso_find_local_bmu(const vso_som *som, const float *v, size_t last_bmu)
{
    size_t L = vso_comparer_len(som->transform, som->depth);
    float *scratch = (float *)malloc((L ? L : 1) * sizeof(float));
    size_t r = find_local_bmu_scratch(som, v, last_bmu, scratch);
    free(scratch);
    return r;
}

/* ---- Som::calculateNeighbourhoodWeight: Som.cpp:949-975 ---- */
double vso_neighbourhood_weight(size_t cx, size_t cy, size_t bx, size_t by, double sigma)
{
    if (sigma > 1.0) {
        double cxd = (double)cx, cyd = (double)cy, bxd = (double)bx, byd = (double)by;
        return exp(-((cxd - bxd) * (cxd - bxd) / 2.0 / sigma / sigma +
                     (cyd - byd) * (cyd - byd) / 2.0 / sigma / sigma));
    } else if (cx == bx && cy == by) {
        return 1.0;
    }
    return 0.0;
}

/* ---- trainBatchSomEpoch phase 1: Som.cpp:762-806 ---- */
void vso_batch_phase1_range(const vso_som *som, const float *X, size_t B,
                            size_t s0, size_t s1, uint64_t *lastbmu, float *sqres,
                            int is_first, int nthreads)
{
    (void)B;
    const size_t J = som->in_len, D = som->depth;
    const size_t L = vso_comparer_len(som->transform, D);
    if (nthreads < 1)
        nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        float *scratch = (float *)malloc((L ? L : 1) * sizeof(float));
#pragma omp for schedule(dynamic, 4)
        for (long long s = (long long)s0; s < (long long)s1; ++s) {
            const float *x = X + (size_t)s * J;
            size_t idx = is_first ? find_bmu_scratch(som, x, scratch)               /* :771 */
                                  : find_local_bmu_scratch(som, x, (size_t)lastbmu[s], scratch); /* :793 */
            lastbmu[s] = idx; /* :777,800 */
            /* residual = Comparer(x, M[idx], ...); residual.squaredNorm()  :780-781 */
            vso_comparer(som->transform, x, J, som->map + idx * D, D, scratch);
            sqres[s] = vso_dot_self(scratch, L);
        }
        free(scratch);
    }
}

/* bmuHits[index] += 1 (:778,801) and meanSquareError += sq / (float)epochSize
 * (:781,804) in sample order (the reference's atomic order is nondeterministic, Q13). */
float vso_batch_phase1_finish(vso_som *som, const uint64_t *lastbmu,
                              const float *sqres, size_t B)
{
    float mse = 0.0f;
    float fB = (float)B;
    for (size_t s = 0; s < B; ++s) {
        som->hits[lastbmu[s]] += 1u;
        float q = sqres[s] / fB;
        mse = mse + q;
    }
    return mse;
}

/* neighbourhood weights as float, tabulated over (|dx|,|dy|): identical values to
 * calling vso_neighbourhood_weight per (node,sample) since the argument depends on
 * (cx-bx)^2 and (cy-by)^2 only (Som.cpp:962-963). */
static float *build_nbh_lut(const vso_som *som, double sigma, size_t *ny_out)
{
    size_t W = som->width, N = som->width * som->height;
    /* largest y SomIndex(som, idx) can produce: (N-W)/H  (Q10) */
    size_t ymax = N ? (N - W) / som->height : 0;
    size_t ny = ymax + 1;
    float *lut = (float *)malloc(ny * W * sizeof(float));
    for (size_t dy = 0; dy < ny; ++dy)
        for (size_t dx = 0; dx < W; ++dx)
            lut[dy * W + dx] = (float)vso_neighbourhood_weight(dx, dy, 0, 0, sigma); /* :851 cast */
    *ny_out = ny;
    return lut;
}

static inline size_t absdiff(size_t a, size_t b) { return a > b ? a - b : b - a; }

/* ---- trainBatchSomEpoch phase 2: Som.cpp:809-876 (Q4, Q7, Q14) ---- */
void vso_batch_phase2_range(vso_som *som, const float *X, size_t B,
                            const uint64_t *lastbmu, double sigma,
                            size_t n0, size_t n1, int nthreads)
{
    const size_t J = som->in_len, D = som->depth, W = som->width;
    const int tr = som->transform;
    if (nthreads < 1)
        nthreads = 1;
    size_t ny;
    float *lut = build_nbh_lut(som, sigma, &ny);
    size_t *bxs = (size_t *)malloc((B ? B : 1) * sizeof(size_t));
    size_t *bys = (size_t *)malloc((B ? B : 1) * sizeof(size_t));
    for (size_t j = 0; j < B; ++j)
        vso_somindex(som, (size_t)lastbmu[j], &bxs[j], &bys[j]); /* :847-849 */

#pragma omp parallel num_threads(nthreads)
    {
        float *M = (float *)malloc((D ? D : 1) * sizeof(float));
        float *S = (float *)malloc((D ? D : 1) * sizeof(float));
        float *delta = (float *)malloc((D ? D : 1) * sizeof(float));
#pragma omp for schedule(dynamic, 4)
        for (long long ni = (long long)n0; ni < (long long)n1; ++ni) {
            size_t node = (size_t)ni, cx, cy;
            vso_somindex(som, node, &cx, &cy); /* :816-820 */
            float sumW = 0.f;                  /* :840 */
            for (size_t d = 0; d < D; ++d) {
                M[d] = 0.f; /* :843 */
                S[d] = 0.f; /* :844 */
            }
            for (size_t j = 0; j < B; ++j) {
                float w = lut[absdiff(cy, bys[j]) * W + absdiff(cx, bxs[j])]; /* :851 */
                sumW = sumW + w;     /* :857 */
                float c = w / sumW;  /* currentWeight / sumOfWeights  :864 */
                const float *x = X + j * J;
                if (tr == VSO_STANDARD) {
                    for (size_t d = 0; d < D; ++d) {
                        float dl = x[d] - M[d]; /* Stepper :861 */
                        float t = c * dl;
                        M[d] = M[d] + t;        /* :864 */
                        float u = w * dl;       /* (w*Stepper(x,lastModel)) :867 */
                        u = u * dl;             /*   .* currentDelta               */
                        S[d] = S[d] + u;
                    }
                } else {
                    vso_stepper(tr, x, J, M, D, delta); /* :861 (== :867's second call) */
                    for (size_t d = 0; d < D; ++d) {
                        float dl = delta[d];
                        float t = c * dl;
                        M[d] = M[d] + t;
                        float u = w * dl;
                        u = u * dl;
                        S[d] = S[d] + u;
                    }
                }
            }
            for (size_t d = 0; d < D; ++d) {
                som->map[node * D + d] = M[d];                   /* :870 */
                som->sigma[node * D + d] = sqrtf(S[d] / sumW);   /* :873 */
            }
            som->weight[node] = sumW; /* :875 */
        }
        free(M);
        free(S);
        free(delta);
    }
    free(lut);
    free(bxs);
    free(bys);
}

float vso_batch_epoch(vso_som *som, const float *X, size_t B, uint64_t *lastbmu,
                      double sigma, int is_first, int nthreads)
{
    float *sq = (float *)malloc((B ? B : 1) * sizeof(float));
    vso_batch_phase1_range(som, X, B, 0, B, lastbmu, sq, is_first, nthreads);
    float mse = vso_batch_phase1_finish(som, lastbmu, sq, B);
    vso_batch_phase2_range(som, X, B, lastbmu, sigma, 0, som->width * som->height, nthreads);
    free(sq);
    return mse;
}

/* ---- Som::trainBatchSom: Som.cpp:716-754 ---- */
size_t vso_train_batch(vso_som *som, const float *X, const size_t *chunk_off,
                       size_t nchunks, size_t epochs, double sigma0,
                       double sigma_decay, float *mse_out, int nthreads)
{
    size_t maxB = 0;
    for (size_t c = 0; c < nchunks; ++c)
        maxB = VSO_MAX(maxB, chunk_off[c + 1] - chunk_off[c]);
    uint64_t *lastbmu = (uint64_t *)malloc((maxB ? maxB : 1) * sizeof(uint64_t));
    size_t done = 0;
    for (size_t i = 0; i < epochs; ++i) {
        double sigma = sigma0 * exp(-sigma_decay * (double)i); /* :727 */
        if (sigma < 1.0)                                       /* :729-730 */
            break;
        float mse = 0.0f;
        size_t count = 0;
        for (size_t c = 0; c < nchunks; ++c) { /* :735-741 */
            size_t B = chunk_off[c + 1] - chunk_off[c];
            memset(lastbmu, 0, B * sizeof(uint64_t)); /* DataSet.cpp:136-137 */
            mse += vso_batch_epoch(som, X + chunk_off[c] * som->in_len, B, lastbmu,
                                   sigma, i == 0, nthreads);
            ++count;
        }
        mse /= (float)count; /* :743 */
        if (mse_out)
            mse_out[i] = mse;
        ++done;
    }
    free(lastbmu);
    return done;
}

/* ---- Som::trainSingle: Som.cpp:885-947 (Q6, Q8, Q11) ---- */
size_t vso_train_single(vso_som *som, const float *v, double eta, double sigma,
                        uint64_t *last_bmu, int decay_fn,
                        float *residual_out, float *dist_out)
{
    const size_t W = som->width, H = som->height, D = som->depth, J = som->in_len;
    const int tr = som->transform;
    const size_t L = vso_comparer_len(tr, D);
    float *scratch = (float *)malloc((VSO_MAX(L, D) ? VSO_MAX(L, D) : 1) * sizeof(float));
    float *delta = (float *)malloc((D ? D : 1) * sizeof(float));
    float *delta2 = (float *)malloc((D ? D : 1) * sizeof(float));

    /* :889-892  SIGMA_SWITCH_TO_LOCAL == 1 */
    size_t bmu = sigma > 1 ? find_bmu_scratch(som, v, scratch)
                           : find_local_bmu_scratch(som, v, (size_t)*last_bmu, scratch);
    size_t bx = bmu % W, by = bmu / W;
    *last_bmu = by * W + bx; /* :895 */

    /* :899-903 truncating window */
    size_t startX = (size_t)fmax((double)bx - 2.5 * sigma, 0.);
    size_t startY = (size_t)fmax((double)by - 2.5 * sigma, 0.);
    size_t endX = (size_t)fmin((double)bx + 2.5 * sigma, (double)W);
    size_t endY = (size_t)fmin((double)by + 2.5 * sigma, (double)H);

    for (size_t j = startY; j < endY; j++) {
        for (size_t i = startX; i < endX; i++) {
            size_t n = j * W + i;
            float *M = som->map + n * D, *S = som->S + n * D, *sg = som->sigma + n * D;
            vso_stepper(tr, v, J, M, D, delta);                            /* :912 */
            double h = vso_neighbourhood_weight(i, j, bx, by, sigma);      /* :915 */
            if (decay_fn == VSO_EXPONENTIAL) {
                som->weight[n] += (float)(h * eta);                        /* :924 */
                float sc = (float)(h * eta); /* double scalar narrowed before the fp32 product */
                for (size_t d = 0; d < D; ++d) {
                    float t = sc * delta[d];
                    M[d] = M[d] + t;                                       /* :925 */
                }
            } else {
                som->weight[n] += (float)h;                                /* :930 */
                double tw = som->weight[n] == 0 ? 1.0 : h / (double)som->weight[n]; /* :933 */
                float sc = (float)tw;
                vso_stepper(tr, v, J, M, D, delta2);                       /* :935 */
                for (size_t d = 0; d < D; ++d) {
                    float t = sc * delta2[d];
                    M[d] = M[d] + t;
                }
            }
            double tw2 = som->weight[n] == 0 ? 0.000001 : (double)som->weight[n]; /* :939 */
            float hf = (float)h, twf = (float)tw2;
            vso_stepper(tr, v, J, M, D, delta2); /* Stepper(v, map_new) :941 */
            for (size_t d = 0; d < D; ++d) {
                float pr = delta[d] * delta2[d];
                float t = hf * pr;
                S[d] = S[d] + t;                           /* :941 */
                sg[d] = sqrtf(fabsf(S[d] / twf));          /* :942 */
            }
        }
    }
    /* :946 */
    vso_comparer(tr, v, J, som->map + bmu * D, D, scratch);
    if (residual_out)
        memcpy(residual_out, scratch, L * sizeof(float));
    if (dist_out)
        *dist_out = vso_dot_self(scratch, L); /* (float)euclidianWeightedDist(bmu,...) */
    free(scratch);
    free(delta);
    free(delta2);
    return bmu;
}

/* ---- inner loop of trainBasicSom: Som.cpp:1159-1171 ---- */
/* One chunk of the sample loop.  The reference keeps ONE running float over all chunks of an epoch
 * (meanSquareError is declared before the chunk loop, Som.cpp:1153, and every sample adds
 * squaredNorm/epochSize of ITS chunk, :1167); mse_start is that running value on entry and the
 * return value is the running value after this chunk. */
float vso_train_online_chunk_from(vso_som *som, const float *X, size_t B, uint64_t *lastbmu,
                                  double eta, double sigma, int decay_fn, float mse_start)
{
    const size_t L = vso_comparer_len(som->transform, som->depth);
    float *res = (float *)malloc((L ? L : 1) * sizeof(float));
    float mse = mse_start, fB = (float)B;
    for (size_t j = 0; j < B; ++j) {
        size_t pos = vso_train_single(som, X + j * som->in_len, eta, sigma, &lastbmu[j],
                                      decay_fn, res, NULL);
        som->hits[pos] += 1u;                 /* addBmu :1165,1189-1192 */
        float q = vso_dot_self(res, L) / fB;  /* residual.squaredNorm()/epochSize :1167 */
        mse = mse + q;
    }
    free(res);
    return mse;
}

float vso_train_online_chunk(vso_som *som, const float *X, size_t B,
                             uint64_t *lastbmu, double eta, double sigma, int decay_fn)
{
    return vso_train_online_chunk_from(som, X, B, lastbmu, eta, sigma, decay_fn, 0.0f);
}

/* ---- Som::trainBasicSom: Som.cpp:1135-1187 ---- */
void vso_train_online(vso_som *som, const float *X, const size_t *chunk_off,
                      size_t nchunks, size_t epochs, double eta0, double eta_decay,
                      double sigma0, double sigma_decay, int decay_fn, float *mse_out)
{
    size_t maxB = 0;
    for (size_t c = 0; c < nchunks; ++c)
        maxB = VSO_MAX(maxB, chunk_off[c + 1] - chunk_off[c]);
    uint64_t *lastbmu = (uint64_t *)malloc((maxB ? maxB : 1) * sizeof(uint64_t));
    for (size_t i = 0; i < epochs; ++i) {
        double eta = eta0 * exp(-eta_decay * (double)i);       /* :1145 */
        double sigma = sigma0 * exp(-sigma_decay * (double)i); /* :1146 */
        if (sigma < 1.0)                                       /* :1148-1149 */
            sigma = 1.0;
        float mse = 0.0f;
        size_t count = 0;
        for (size_t c = 0; c < nchunks; ++c) {
            size_t B = chunk_off[c + 1] - chunk_off[c];
            memset(lastbmu, 0, B * sizeof(uint64_t)); /* DataSet.cpp:136-137 */
            mse = vso_train_online_chunk_from(som, X + chunk_off[c] * som->in_len, B, lastbmu,
                                              eta, sigma, decay_fn, mse); /* one accumulator :1153,1167 */
            ++count;
        }
        mse /= (float)count; /* :1175 */
        if (mse_out)
            mse_out[i] = mse;
    }
    free(lastbmu);
}

/* ---- timing-honest single-thread variant: same arithmetic as vso_batch_epoch, but
 * with the reference's allocation pattern per call: neuron copy (Som.cpp:136,199-202),
 * a fresh heap vector per Comparer/Stepper result (std::function returning VectorXf),
 * lastModel copy (:859), and one libm exp per (node,sample) (:851).  ---- */
static float *heap_copy(const float *src, size_t n)
{
    float *p = (float *)malloc((n ? n : 1) * sizeof(float));
    memcpy(p, src, n * sizeof(float));
    return p;
}

static double dist_faithful(const vso_som *som, size_t node, const float *v)
{
    size_t D = som->depth, L = vso_comparer_len(som->transform, D);
    float *sM = (float *)malloc((D ? D : 1) * sizeof(float)); /* :128,131 select */
    for (size_t d = 0; d < D; ++d)
        sM[d] = som->sigma[node * D + d] < 0.00001f ? 0.00001f : som->sigma[node * D + d];
    float *validEigen = (float *)malloc((som->in_len ? som->in_len : 1) * sizeof(float)); /* :134 */
    for (size_t d = 0; d < som->in_len; ++d)
        validEigen[d] = 1.0f * 1.0f;
    float *neuron = heap_copy(som->map + node * D, D); /* getNeuron copy :136 */
    float *res = (float *)malloc((L ? L : 1) * sizeof(float));
    vso_comparer(som->transform, v, som->in_len, neuron, D, res);
    float r = vso_dot_self(res, L);
    free(res);
    free(neuron);
    free(validEigen);
    free(sM);
    return (double)r;
}

float vso_batch_epoch_faithful(vso_som *som, const float *X, size_t B,
                               uint64_t *lastbmu, double sigma, int is_first)
{
    const size_t J = som->in_len, D = som->depth, N = som->width * som->height;
    const size_t L = vso_comparer_len(som->transform, D);
    float mse = 0.0f;
    for (size_t s = 0; s < B; ++s) {
        const float *x = X + s * J;
        size_t idx;
        if (is_first) {
            double minDist = dist_faithful(som, 0, x);
            idx = 0;
            for (size_t i = 0; i < N; ++i) {
                double cur = dist_faithful(som, i, x);
                if (cur < minDist) {
                    minDist = cur;
                    idx = i;
                }
            }
        } else {
            idx = vso_find_local_bmu(som, x, (size_t)lastbmu[s]);
        }
        lastbmu[s] = idx;
        som->hits[idx] += 1u;
        float *neuron = heap_copy(som->map + idx * D, D);
        float *res = (float *)malloc((L ? L : 1) * sizeof(float));
        vso_comparer(som->transform, x, J, neuron, D, res);
        float q = vso_dot_self(res, L) / (float)B;
        mse = mse + q;
        free(res);
        free(neuron);
    }
    float *M = (float *)malloc((D ? D : 1) * sizeof(float));
    float *S = (float *)malloc((D ? D : 1) * sizeof(float));
    for (size_t node = 0; node < N; ++node) {
        size_t cx, cy;
        vso_somindex(som, node, &cx, &cy);
        uint64_t *bmus = (uint64_t *)malloc((B ? B : 1) * sizeof(uint64_t)); /* getLastBMU copy :822 */
        memcpy(bmus, lastbmu, B * sizeof(uint64_t));
        float sumW = 0.f;
        memset(M, 0, D * sizeof(float));
        memset(S, 0, D * sizeof(float));
        for (size_t j = 0; j < B; ++j) {
            size_t bx, by;
            vso_somindex(som, (size_t)bmus[j], &bx, &by);
            float w = (float)vso_neighbourhood_weight(cx, cy, bx, by, sigma);
            sumW = sumW + w;
            float *lastModel = heap_copy(M, D);                  /* :859 */
            float *delta = (float *)malloc((D ? D : 1) * sizeof(float));
            vso_stepper(som->transform, X + j * J, J, M, D, delta); /* :861 */
            float c = w / sumW;
            for (size_t d = 0; d < D; ++d) {
                float t = c * delta[d];
                M[d] = M[d] + t;
            }
            float *delta2 = (float *)malloc((D ? D : 1) * sizeof(float));
            vso_stepper(som->transform, X + j * J, J, lastModel, D, delta2); /* :867 */
            for (size_t d = 0; d < D; ++d) {
                float u = w * delta2[d];
                u = u * delta[d];
                S[d] = S[d] + u;
            }
            free(delta2);
            free(delta);
            free(lastModel);
        }
        for (size_t d = 0; d < D; ++d) {
            som->map[node * D + d] = M[d];
            som->sigma[node * D + d] = sqrtf(S[d] / sumW);
        }
        som->weight[node] = sumW;
        free(bmus);
    }
    free(M);
    free(S);
    return mse;
}

/* ================= "next" rows ================= */

/* ---- Som::findRestrictedBmu: Som.cpp:313-332 ---- */
size_t vso_find_restricted_bmu(const vso_som *som, const float *v, uint64_t min_hits)
{
    size_t N = som->width * som->height;
    size_t L = vso_comparer_len(som->transform, som->depth);
    float *scratch = (float *)malloc((L ? L : 1) * sizeof(float));
    double minDist = (double)dist_scratch(som, 0, v, scratch); /* node 0 seeds whatever its hits :316-317 */
    size_t minIndex = 0;
    for (size_t i = 0; i < N; ++i) {
        double cur = (double)dist_scratch(som, i, v, scratch);
        if (cur < minDist && som->hits[i] >= min_hits) { /* :322 */
            minDist = cur;
            minIndex = i;
        }
    }
    free(scratch);
    return minIndex;
}

/* ---- Som::findRestrictedBmd: Som.cpp:457-487 ---- */
void vso_find_restricted_bmd(const vso_som *som, const float *v, uint64_t min_hits, double *out)
{
    size_t N = som->width * som->height;
    size_t L = vso_comparer_len(som->transform, som->depth);
    float *scratch = (float *)malloc((L ? L : 1) * sizeof(float));
    double C = 0;
    for (size_t i = 0; i < N; ++i) {
        if (som->hits[i] >= min_hits) {
            double d = (double)dist_scratch(som, i, v, scratch); /* :470 */
            d = exp(-d * d / 2);                                  /* :473 */
            out[i] = d;
            C += d;                                               /* :476 */
        } else {
            out[i] = 0;
        }
    }
    for (size_t i = 0; i < N; ++i)
        out[i] /= C; /* :483-484 */
    free(scratch);
}

/* ---- Som::euclidianWeightedDistRaw: Som.cpp:143-157 (valid = weights = 1) ----
 * a = (M - v)/sM ; b = ((M - v) * validEigen)/sM ; a.dot(b) in Eigen's redux order (Q1). */
double vso_dist_raw(const vso_som *som, size_t pos, const float *v)
{
    size_t D = som->depth;
    const float *M = som->map + pos * D, *sg = som->sigma + pos * D;
    /* the products a_d*b_d are summed in the order of vso_dot_self; reuse its structure */
    float *prod = (float *)malloc((D ? D : 1) * sizeof(float));
    for (size_t d = 0; d < D; ++d) {
        float sM = sg[d] < 0.00001f ? 0.00001f : sg[d]; /* :150 */
        float r = M[d] - v[d];
        float a = r / sM;
        float rb = r * (1.0f * 1.0f);                   /* validEigen = valid*weights :153 */
        float b = rb / sM;
        prod[d] = a * b;
    }
    /* Eigen redux over the products (same tree as vso_dot_self, without squaring) */
    float res;
    size_t n = D, aligned2 = (n / 8) * 8, aligned = (n / 4) * 4;
    if (n == 0) {
        res = 0.f;
    } else if (aligned) {
        float p0[4], p1[4];
        for (int k = 0; k < 4; ++k)
            p0[k] = prod[k];
        if (aligned > 4) {
            for (int k = 0; k < 4; ++k)
                p1[k] = prod[4 + k];
            for (size_t idx = 8; idx < aligned2; idx += 8) {
                for (int k = 0; k < 4; ++k)
                    p0[k] = p0[k] + prod[idx + k];
                for (int k = 0; k < 4; ++k)
                    p1[k] = p1[k] + prod[idx + 4 + k];
            }
            for (int k = 0; k < 4; ++k)
                p0[k] = p0[k] + p1[k];
            if (aligned > aligned2)
                for (int k = 0; k < 4; ++k)
                    p0[k] = p0[k] + prod[aligned2 + k];
        }
        float t02 = p0[0] + p0[2], t13 = p0[1] + p0[3];
        res = t02 + t13;
        for (size_t idx = aligned; idx < n; ++idx)
            res = res + prod[idx];
    } else {
        res = prod[0];
        for (size_t idx = 1; idx < n; ++idx)
            res = res + prod[idx];
    }
    free(prod);
    return (double)res;
}

/* ---- Som::updateUMatrix: Som.cpp:999-1111 ---- */
void vso_update_umatrix(const vso_som *som, double *U)
{
    const size_t width = som->width, height = som->height, D = som->depth;
    const double diagonalFactor = 0.3;
#define RAW(ii, jj, ni, nj) vso_dist_raw(som, (ii) * width + (jj), som->map + ((ni) * width + (nj)) * D)
    for (size_t i = 0; i < height; ++i) {
        for (size_t j = 0; j < width; ++j) {
            double u;
            if (j > 0 && i > 0 && j < (width - 1) && i < (height - 1)) {
                u = (RAW(i, j, i, j - 1) + RAW(i, j, i, j + 1) + RAW(i, j, i + 1, j) + RAW(i, j, i - 1, j) +
                     RAW(i, j, i - 1, j - 1) * diagonalFactor + RAW(i, j, i + 1, j - 1) * diagonalFactor +
                     RAW(i, j, i - 1, j + 1) * diagonalFactor + RAW(i, j, i + 1, j + 1) * diagonalFactor) / 8;
            } else if (i == 0 && j > 0 && j < (width - 1)) {
                u = (RAW(i, j, i, j - 1) + RAW(i, j, i, j + 1) + RAW(i, j, i + 1, j) +
                     RAW(i, j, i + 1, j - 1) * diagonalFactor + RAW(i, j, i + 1, j + 1) * diagonalFactor) / 5;
            } else if (i == (height - 1) && j > 0 && j < (width - 1)) {
                u = (RAW(i, j, i, j - 1) + RAW(i, j, i, j + 1) + RAW(i, j, i - 1, j) +
                     RAW(i, j, i - 1, j - 1) * diagonalFactor + RAW(i, j, i - 1, j + 1) * diagonalFactor) / 5;
            } else if (j == 0 && i > 0 && i < (height - 1)) {
                u = (RAW(i, j, i, j + 1) + RAW(i, j, i + 1, j) + RAW(i, j, i - 1, j) +
                     RAW(i, j, i - 1, j + 1) * diagonalFactor + RAW(i, j, i + 1, j + 1) * diagonalFactor) / 5;
            } else if (j == (width - 1) && i > 0 && i < (height - 1)) {
                u = (RAW(i, j, i, j - 1) + RAW(i, j, i + 1, j) + RAW(i, j, i - 1, j) +
                     RAW(i, j, i - 1, j - 1) * diagonalFactor + RAW(i, j, i + 1, j - 1) * diagonalFactor) / 5;
            } else if (j == 0 && i == 0) {
                u = (RAW(i, j, i, j + 1) + RAW(i, j, i + 1, j) + RAW(i, j, i + 1, j + 1) * diagonalFactor) / 3;
            } else if (j == (width - 1) && i == 0) {
                u = (RAW(i, j, i, j - 1) + RAW(i, j, i + 1, j) + RAW(i, j, i + 1, j - 1) * diagonalFactor) / 3;
            } else if (j == 0 && i == (height - 1)) {
                u = (RAW(i, j, i, j + 1) + RAW(i, j, i - 1, j) + RAW(i, j, i - 1, j + 1) * diagonalFactor) / 3;
            } else if (j == (width - 1) && i == (height - 1)) {
                u = (RAW(i, j, i, j - 1) + RAW(i, j, i - 1, j) + RAW(i, j, i - 1, j - 1) * diagonalFactor) / 3;
            } else {
                u = 0;
            }
            U[i * width + j] = u;
        }
    }
#undef RAW
}
