// asan_driver.cpp -- runs the GPU-free C side under AddressSanitizer + UndefinedBehaviorSanitizer (tools/sanitize/Makefile).
// Exact-size heap buffers everywhere, so that a read or write one element past an operand is a report, not luck.
#include "../../navigation-by-deja-vu_amd/csrc/dejavu_host.inl"

#include <cstdio>
#include <cstdlib>
#include <vector>

extern "C" {
void oracle_sads_hsv(const uint8_t* library, int64_t F, int h, int w, const uint8_t* scene, double chem_weight, double* fambuf);
int oracle_step(const uint8_t* library, int64_t F, int h, int w, const uint8_t* patches, int A, double chem_weight,
                double* angle_fam, double* scene_fam, int32_t* best_idex, int64_t* best_view, double* step_fam);
double oracle_ssds(const double* a, const double* b, int64_t n, int64_t m);
void oracle_int_sums(const uint8_t* library, int64_t F, int h, int w, const uint8_t* scene, int64_t* s_hs, int64_t* s_v);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
#define REQUIRE(x) do { if (!(x)) { std::fprintf(stderr, "asan_driver: check failed: %s (line %d)\n", #x, __LINE__); std::exit(2); } } while (0)

static void oracle_cases() {
    const int shapes[][4] = {{1, 1, 1, 1}, {3, 5, 7, 2}, {65, 8, 8, 8}, {130, 9, 31, 5}, {17, 1, 64, 64}};
    for (const auto& s : shapes) {
        const int64_t F = s[0];
        const int h = s[1], w = s[2], A = s[3];
        const size_t P = (size_t)h * w;
        uint8_t* lib = (uint8_t*)std::malloc((size_t)F * P * 3);
        uint8_t* pat = (uint8_t*)std::malloc((size_t)A * P * 3);
        for (size_t i = 0; i < (size_t)F * P * 3; ++i) lib[i] = (uint8_t)(rnd() % 5 * 63);
        for (size_t i = 0; i < (size_t)A * P * 3; ++i) pat[i] = (uint8_t)(rnd() % 5 * 63);
        std::memcpy(pat, lib + (size_t)(F / 2) * P * 3, P * 3);                   // heading 0 stands on a stored view
        for (double cw : {0.0, 0.25, 1.0}) {
            double* fam = (double*)std::malloc(sizeof(double) * F);
            oracle_sads_hsv(lib, F, h, w, pat, cw, fam);
            REQUIRE(fam[F / 2] == (double)P);
            std::free(fam);
            double* ang = (double*)std::malloc(sizeof(double) * A);
            double* scene = (double*)std::malloc(sizeof(double) * F);
            int32_t best = -1;
            int64_t view = -1;
            double sf = 0;
            REQUIRE(oracle_step(lib, F, h, w, pat, A, cw, ang, scene, &best, &view, &sf) == 0);
            REQUIRE(best == 0 && sf == (double)P && ang[0] == (double)P);
            REQUIRE(oracle_step(lib, F, h, w, pat, A, cw, ang, nullptr, &best, &view, &sf) == 0);
            std::free(ang);
            std::free(scene);
        }
        int64_t* shs = (int64_t*)std::malloc(sizeof(int64_t) * F);
        int64_t* sv = (int64_t*)std::malloc(sizeof(int64_t) * F);
        oracle_int_sums(lib, F, h, w, pat, shs, sv);
        REQUIRE(shs[F / 2] == 0 && sv[F / 2] == 0);
        std::free(shs);
        std::free(sv);
        std::free(lib);
        std::free(pat);
    }
    double* a = (double*)std::malloc(sizeof(double) * 35);
    double* b = (double*)std::malloc(sizeof(double) * 35);
    for (int i = 0; i < 35; ++i) { a[i] = (double)(rnd() % 100) / 7.0; b[i] = a[i] + 1.0; }
    REQUIRE(oracle_ssds(a, b, 5, 7) == 35.0);
    std::free(a);
    std::free(b);
}

static void merge_cases() {
    for (int it = 0; it < 2000; ++it) {
        const int world = 1 + (int)(rnd() % 64), A = 1 + (int)(rnd() % DV_MAX_HEADINGS);
        const int64_t stride = 3 + 4 * (int64_t)A;                                 // exact: no slack behind a record
        double* rec = (double*)std::malloc(sizeof(double) * (size_t)world * stride);
        uint64_t* keys = (uint64_t*)std::malloc(sizeof(uint64_t) * (size_t)(A + 4 * world));
        std::memset(keys, 0, sizeof(uint64_t) * (size_t)(A + 4 * world));
        for (int r = 0; r < world; ++r) {
            double* q = rec + (size_t)r * stride;
            double mx = -1e300;
            int arg = 0;
            for (int a = 0; a < A; ++a) {
                q[3 + a] = (double)(rnd() % 7) + ((it & 1) ? 0.0 : (double)(rnd() % 1000) * 1e-13);
                if (q[3 + a] > mx) { mx = q[3 + a]; arg = a; }
                q[3 + A + a] = (double)(rnd() % (1ull << 33));
                q[3 + 2 * A + a] = q[3 + a];
                q[3 + 3 * A + a] = q[3 + A + a];
            }
            q[0] = mx;
            q[1] = (double)(1 + rnd() % 3);
            q[2] = (double)(rnd() % 3);
            auto key = [](double d) { uint64_t v; std::memcpy(&v, &d, 8); return (v >> 63) ? ~v : (v | 0x8000000000000000ull); };
            for (int a = 0; a < A; ++a) { const uint64_t k = key(q[3 + a]); if (k > keys[a]) keys[a] = k; }
            uint64_t* slot = keys + A + 4 * r;
            slot[0] = key(q[0]);
            slot[1] = (uint64_t)q[1] | ((uint64_t)q[2] << 32) | (1ull << 48);
            slot[2] = (uint64_t)(arg + 1);
            slot[3] = (uint64_t)q[3 + A + arg] + 1;
        }
        dv_merge_out out;
        const int rc = dv_merge_records(rec, world, A, stride, 1e-9, &out);
        REQUIRE(rc == DV_OK);
        if (!out.needs_resolve) REQUIRE(out.best_heading >= 0 && out.best_heading < A);
        dv_merge_out ko;
        REQUIRE(dv_merge_keys(keys, world, A, 1e-9, 0, &ko) == DV_OK);
        if (!ko.needs_resolve && !out.needs_resolve && !out.resolved) {
            REQUIRE(ko.best_heading == out.best_heading && ko.best_view == out.best_view && ko.best_fam == out.best_fam);
        }
        for (int i = 0; i < A + 4 * world; ++i) keys[i] ^= 0x8000000000000000ull;
        dv_merge_out ks;
        REQUIRE(dv_merge_keys(keys, world, A, 1e-9, 1, &ks) == DV_OK);
        REQUIRE(ks.needs_resolve == ko.needs_resolve && ks.best_heading == ko.best_heading);
        rec[2] += 4.0;                                                              // sensed past the landscape
        REQUIRE(dv_merge_records(rec, world, A, stride, 1e-9, &out) == DV_ERR_INDEX);
        REQUIRE(dv_merge_records(rec, 0, A, stride, 1e-9, &out) == DV_ERR_INVALID);
        REQUIRE(dv_merge_keys(keys, world, 0, 1e-9, 0, &ko) == DV_ERR_INVALID);
        std::free(rec);
        std::free(keys);
    }
}

static void plan_cases() {
    for (int it = 0; it < 2000; ++it) {
        uint32_t presence[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int n = 1 + (int)(rnd() % 12);
        for (int i = 0; i < n; ++i) { const int v = (int)(rnd() % 256); presence[v >> 5] |= 1u << (v & 31); }
        const int cap = (int)(rnd() % 20);
        uint8_t* lo = (uint8_t*)std::malloc(cap ? cap : 1);
        uint8_t* w = (uint8_t*)std::malloc(cap ? cap : 1);
        int lmin = -1, lmax = -1;
        const int t = dv_bitplane_plan(presence, cap, lo, w, &lmin, &lmax);
        REQUIRE(t >= -1 && t <= cap);
        if (t > 0) {
            int at = lmin;
            for (int k = 0; k < t; ++k) { REQUIRE(lo[k] == at && w[k] >= 1 && w[k] <= 127); at += w[k]; }
            REQUIRE(at == lmax);
        }
        if (t >= 0) {                                          // fp4 plan of the same planes: exactly-sized buffers
            uint8_t* wfull = (uint8_t*)std::malloc(t ? t : 1);
            int wacc[4] = {-1, -1, -1, -1};
            const int ok = dv_fp4_plan(presence, t, lo, w, wfull, wacc);
            REQUIRE(ok == 0 || ok == 1);
            int span = 0;
            for (int k = 0; k < t; ++k) { REQUIRE(wfull[k] == 0 || wfull[k] >= w[k]); span += wfull[k]; }
            REQUIRE(span == lmax - lmin);                      // the first planes of the gaps cover the level range once
            std::free(wfull);
        }
        std::free(lo);
        std::free(w);
    }
}

int main() {
    oracle_cases();
    merge_cases();
    plan_cases();
    std::puts("asan_driver: ok (oracle C, dv_merge_records, dv_merge_keys, dv_bitplane_plan, dv_fp4_plan under ASan + UBSan)");
    return 0;
}
