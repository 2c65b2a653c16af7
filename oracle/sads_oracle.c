/*
 * oracle/sads_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the scene-familiarity hot path of the reference
 * (navsim).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library; the shipped path (navigation-by-deja-vu_amd/)
 * never links, imports or calls it.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against
 * outputs of the reference's own code (Cython util.pyx built in /tmp and the
 * agent class imported from it) by tests/test_oracle_golden.py, using the
 * fixtures under tests/golden/ made by tests/golden/make_golden.py.
 *
 * Build: gcc -O2 -ffp-contract=off (no FMA contraction: the reference's
 * extension is built by setuptools with plain -O2 and no -march, so every
 * double operation is individually rounded).  See oracle/Makefile.
 *
 * Citations are into /root/reference/.
 */
#include <stdint.h>
#include <stdlib.h>
#include <math.h>

/*
 * navsim/util.pyx:31-73  sads_hsv_metric
 *
 * For each stored view f (loop order f -> row i -> col j, strictly sequential
 * double accumulation, util.pyx:44-72):
 *   hue equal  : thispx = abs(S_s - S_f)           (util.pyx:48-50; uint8
 *   hue differs: thispx = S_s + S_f                 operands promote to int)
 *   thispx *= 0.5 ; thispx *= chem_weight           (util.pyx:59,68)
 *   thispx += (1 - chem_weight) * abs(V_s - V_f)    (util.pyx:69)
 *   thispx /= 255. ; diff += thispx                 (util.pyx:71-72)
 * fambuf[f] = (double)(h*w) - diff                  (util.pyx:42,73)
 *
 * library: uint8[F,h,w,3] C-contiguous, channels H,S,V (NavBySceneFamiliarity.py:122)
 * scene  : uint8[h,w,3]
 */
void oracle_sads_hsv(const uint8_t *library, int64_t F, int h, int w,
                     const uint8_t *scene, double chem_weight, double *fambuf)
{
    const int64_t npx = (int64_t)h * (int64_t)w;
    const double maxfam = (double)npx;
    for (int64_t f = 0; f < F; ++f) {
        const uint8_t *view = library + f * npx * 3;
        double diff = 0.0;
        for (int i = 0; i < h; ++i) {
            for (int j = 0; j < w; ++j) {
                const int64_t o = ((int64_t)i * w + j) * 3;
                double thispx;
                if (scene[o + 0] == view[o + 0])
                    thispx = (double)abs((int)scene[o + 1] - (int)view[o + 1]);
                else
                    thispx = (double)((int)scene[o + 1] + (int)view[o + 1]);
                thispx *= 0.5;
                thispx *= chem_weight;
                thispx += (1 - chem_weight) * (double)abs((int)scene[o + 2] - (int)view[o + 2]);
                thispx /= 255.;
                diff += thispx;
            }
        }
        fambuf[f] = maxfam - diff;
    }
}

/*
 * navsim/NavBySceneFamiliarity.py:283-316  heading loop of step_forward,
 * given the A sensor patches that get_sensor_mat would have produced (:293).
 *
 *   scene_familiarity[:] = +inf                                  (:287)
 *   for a: temp_fam = kernel(patch[a])                           (:299)
 *          scene_familiarity[f] = min(scene_familiarity[f], temp_fam[f])
 *                  (strict '<' update, :301-303)
 *          angle_familiarity[a] = max_f temp_fam[f]              (:313)
 *   best_idex = argmax(angle_familiarity)  (first maximum, :315)
 *   step_familiarity = angle_familiarity[best_idex]              (:316)
 *
 * best_view is the build's extension (the reference never materialises it):
 * argmax_f temp_fam[f] for the chosen heading, first maximum, i.e. what
 * np.argmax would return on the reference's own fambuf.
 *
 * scene_fam may be NULL.  Returns 0, or -1 if the scratch allocation fails.
 */
int oracle_step(const uint8_t *library, int64_t F, int h, int w,
                const uint8_t *patches, int A, double chem_weight,
                double *angle_fam, double *scene_fam,
                int32_t *best_idex, int64_t *best_view, double *step_fam)
{
    const int64_t npx = (int64_t)h * (int64_t)w;
    double *temp = (double *)malloc(sizeof(double) * (size_t)(F > 0 ? F : 1));
    int64_t *argf = (int64_t *)malloc(sizeof(int64_t) * (size_t)(A > 0 ? A : 1));
    if (!temp || !argf) { free(temp); free(argf); return -1; }

    if (scene_fam)
        for (int64_t f = 0; f < F; ++f) scene_fam[f] = INFINITY;

    for (int a = 0; a < A; ++a) {
        oracle_sads_hsv(library, F, h, w, patches + (int64_t)a * npx * 3, chem_weight, temp);
        double best = -INFINITY;
        int64_t bf = 0;
        for (int64_t f = 0; f < F; ++f) {
            if (scene_fam && temp[f] < scene_fam[f]) scene_fam[f] = temp[f];
            if (temp[f] > best) { best = temp[f]; bf = f; }
        }
        angle_fam[a] = best;
        argf[a] = bf;
    }
    int32_t bi = 0;
    for (int a = 1; a < A; ++a)
        if (angle_fam[a] > angle_fam[bi]) bi = a;
    *best_idex = bi;
    *best_view = argf[bi];
    *step_fam = angle_fam[bi];
    free(temp);
    free(argf);
    return 0;
}

/*
 * navsim/util.pyx:171-184  ssds  (dead code in the reference; the only
 * reference definition of "SSD"): sum over i,j of (a[i,j]-b[i,j])**2, double,
 * row-major, sequential.  Cython lowers `x**2` on a C double to pow(x, 2.0),
 * which glibc evaluates exactly as x*x.
 */
double oracle_ssds(const double *a, const double *b, int64_t n, int64_t m)
{
    double diff = 0.0;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t j = 0; j < m; ++j) {
            const double d = a[i * m + j] - b[i * m + j];
            diff += d * d;
        }
    return diff;
}

/*
 * Exact integer sums behind one view comparison (used by tests to check the
 * HIP kernel's integer accumulators directly): S_hs = sum of the hue-aware
 * saturation term (util.pyx:48-56), S_v = sum |V_s - V_f| (util.pyx:69).
 */
void oracle_int_sums(const uint8_t *library, int64_t F, int h, int w,
                     const uint8_t *scene, int64_t *s_hs, int64_t *s_v)
{
    const int64_t npx = (int64_t)h * (int64_t)w;
    for (int64_t f = 0; f < F; ++f) {
        const uint8_t *view = library + f * npx * 3;
        int64_t hs = 0, v = 0;
        for (int64_t p = 0; p < npx; ++p) {
            const int64_t o = p * 3;
            if (scene[o] == view[o]) hs += abs((int)scene[o + 1] - (int)view[o + 1]);
            else hs += (int)scene[o + 1] + (int)view[o + 1];
            v += abs((int)scene[o + 2] - (int)view[o + 2]);
        }
        s_hs[f] = hs;
        s_v[f] = v;
    }
}

/*
 * Multi-core integer form of the step, for bench.py's second CPU figure only (SURVEY.md section 8d asks for a
 * "fast integer variant with OpenMP on all host cores" beside the reference-equivalent one): the same per-pixel
 * terms as oracle_int_sums (util.pyx:48-56,69), summed as integers, views spread over `threads` OpenMP threads,
 * fam = P - (0.5*cw*S_hs + (1-cw)*S_v)/255.  Not order-exact in the reference's double sense (differences of a few
 * ulp, as for the HIP fast path); the decision rule on top is np.max / np.argmax as in oracle_step.
 * Built in a separate object (liboracle_omp.so, -O3 -fopenmp) so that the pinned liboracle.so keeps its -O2 build.
 */
#ifdef ORACLE_WITH_OPENMP
#include <omp.h>

/*
 * Dense check of a full-size SYNTHETIC library (tests/test_gpu_parity.py:test_dense_oracle_at_config_two): the integer sums
 * of oracle_int_sums (util.pyx:48-56,69) for ONE scene against views [first, first + F) of the build's own synthetic
 * library, each view regenerated on the fly from (seed, view index) -- the counter hash of navsim_amd/synth.py:synth_views
 * (splitmix64 finaliser; the build's generator, no reference code) -- so that the 24.6 GB of a 500 000-view library of
 * 128x128 sensors never exist in host memory.  tests/test_oracle_golden.py pins it to oracle_int_sums on synth_views output.
 */
static inline uint64_t synth_mix(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int oracle_synth_int_sums(uint64_t seed, int64_t first, int64_t F, int h, int w, int full_range_s,
                          const uint8_t *scene, int threads, int64_t *s_hs, int64_t *s_v)
{
    static const int levels[5] = {0, 63, 127, 191, 255};
    const int64_t npx = (int64_t)h * (int64_t)w;
    if (threads < 1) threads = 1;
#pragma omp parallel for schedule(static) num_threads(threads)
    for (int64_t f = 0; f < F; ++f) {
        const uint64_t base = (uint64_t)(first + f) * (uint64_t)npx + seed * 0x9E3779B97F4A7C15ull;
        int64_t hs = 0, v = 0;
        for (int64_t p = 0; p < npx; ++p) {
            const uint64_t z = synth_mix(base + (uint64_t)p);
            const int V = levels[((z & 0xFFFFull) * 5ull) >> 16];
            const int H = (int)((z >> 16) & 1ull) * 127;
            const int S = full_range_s ? (int)((z >> 17) & 0x7Full) : (int)((z >> 17) & 1ull) * 127;
            const int64_t o = p * 3;
            hs += ((int)scene[o] == H) ? abs((int)scene[o + 1] - S) : (int)scene[o + 1] + S;
            v += abs((int)scene[o + 2] - V);
        }
        s_hs[f] = hs;
        s_v[f] = v;
    }
    return threads;
}

int oracle_step_fast(const uint8_t *library, int64_t F, int h, int w, const uint8_t *patches, int A,
                     double cw, int threads, double *angle_fam, int64_t *best_view, int32_t *best_heading)
{
    const int64_t npx = (int64_t)h * (int64_t)w;
    const double whs = 0.5 * cw, wv = 1.0 - cw;
    if (threads < 1) threads = 1;
    for (int a = 0; a < A; ++a) {
        const uint8_t *scene = patches + (int64_t)a * npx * 3;
        double best = -1.0 / 0.0;
        int64_t bestf = -1;
#pragma omp parallel num_threads(threads)
        {
            double lbest = -1.0 / 0.0;
            int64_t lf = -1;
#pragma omp for schedule(static) nowait
            for (int64_t f = 0; f < F; ++f) {
                const uint8_t *view = library + f * npx * 3;
                int64_t hs = 0, v = 0;
                for (int64_t p = 0; p < npx; ++p) {
                    const int64_t o = p * 3;
                    const int ss = scene[o + 1], sf = view[o + 1];
                    hs += (scene[o] == view[o]) ? abs(ss - sf) : ss + sf;
                    v += abs((int)scene[o + 2] - (int)view[o + 2]);
                }
                const double fam = (double)npx - (whs * (double)hs + wv * (double)v) / 255.0;
                if (fam > lbest) { lbest = fam; lf = f; }
            }
#pragma omp critical
            if (lbest > best || (lbest == best && lf >= 0 && (bestf < 0 || lf < bestf))) { best = lbest; bestf = lf; }
        }
        angle_fam[a] = best;
        best_view[a] = bestf;
    }
    int bh = 0;
    for (int a = 1; a < A; ++a) if (angle_fam[a] > angle_fam[bh]) bh = a;
    *best_heading = bh;
    return omp_get_max_threads();
}
#endif
