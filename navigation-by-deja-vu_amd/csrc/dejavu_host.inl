// dejavu_host.inl -- the entry points of include/dejavu.h that are host arithmetic only (no context, no HIP call):
// the sharded decision from gathered records or reduced keys, and the bit-plane plan of a byte plane.  Included by
// dejavu_hip.hip; also compiled on its own, with AddressSanitizer + UndefinedBehaviorSanitizer, by
// tools/sanitize/Makefile, whose driver the CPU test suite runs (GPU sanitizers are not available on the pool).
#include "../../include/dejavu.h"

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>

#ifndef DV_HOST_CONSTANTS
#define DV_HOST_CONSTANTS
namespace dv_host {
constexpr int kMaxHeadings = DV_MAX_HEADINGS;
constexpr int kKeyWordsPerRank = 4;
}
#endif
// Host arithmetic of the sharded decision; mirrors navsim_amd/sharded.py (needs_resolve + merge_records), which the
// CPU tests compare it with on random records.
extern "C" int dv_merge_records(const double* rec, int world, int A, int64_t stride, double delta, dv_merge_out* out) {
    if (!rec || !out || world < 1 || world > 64 || A < 1 || A > dv_host::kMaxHeadings || stride < 3 + 4 * (int64_t)A) return DV_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    // state word: 0..2, + 4 when that rank's patches were sensed past the end of the landscape
    for (int r = 0; r < world; ++r)
        if (rec[(int64_t)r * stride + 2] >= 4.0) return DV_ERR_INDEX;
    auto R = [&](int r, int64_t i) { return rec[(int64_t)r * stride + i]; };
    double gmax = R(0, 0);
    for (int r = 1; r < world; ++r) if (R(r, 0) > gmax) gmax = R(r, 0);
    long long total = 0;
    bool unresolved = false;
    for (int r = 0; r < world; ++r) {
        if (R(r, 0) >= gmax - delta) {
            out->contending_mask |= 1ull << r;
            out->n_contending++;
            total += (long long)R(r, 1);
            if (R(r, 2) == 0.0) unresolved = true;
        }
    }
    // per-heading maxima of the integer-sum scores over ALL ranks, first rank on ties
    int owner[dv_host::kMaxHeadings];
    for (int a = 0; a < A; ++a) {
        double m = R(0, 3 + a);
        int o = 0;
        for (int r = 1; r < world; ++r) if (R(r, 3 + a) > m) { m = R(r, 3 + a); o = r; }
        out->angle_fam[a] = m;
        owner[a] = o;
    }
    if (total > 1 && unresolved) {
        out->needs_resolve = 1;
        return DV_OK;
    }
    if (total <= 1) {
        int best = 0;
        for (int a = 1; a < A; ++a) if (out->angle_fam[a] > out->angle_fam[best]) best = a;     // first maximum
        out->best_heading = best;
        out->best_view = (int64_t)R(owner[best], 3 + A + best);
        out->best_fam = out->angle_fam[best];
        return DV_OK;
    }
    const double ninf = -std::numeric_limits<double>::infinity();
    double ex_a[dv_host::kMaxHeadings];
    for (int a = 0; a < A; ++a) {
        double m = ninf;
        for (int r = 0; r < world; ++r) {
            if (!((out->contending_mask >> r) & 1)) continue;
            const double v = (R(r, 2) == 2.0) ? R(r, 3 + a) : R(r, 3 + 2 * A + a);
            if (v > m) m = v;
        }
        ex_a[a] = m;
    }
    int best = 0;
    for (int a = 1; a < A; ++a) if (ex_a[a] > ex_a[best]) best = a;                               // first maximum
    int64_t best_view = -1;
    bool have = false;
    for (int r = 0; r < world; ++r) {
        if (!((out->contending_mask >> r) & 1)) continue;
        const bool all_exact = R(r, 2) == 2.0;
        const double v = all_exact ? R(r, 3 + best) : R(r, 3 + 2 * A + best);
        if (v == ex_a[best]) {
            const int64_t f = (int64_t)(all_exact ? R(r, 3 + A + best) : R(r, 3 + 3 * A + best));
            if (!have || f < best_view) { best_view = f; have = true; }
        }
    }
    for (int a = 0; a < A; ++a) if (std::isfinite(ex_a[a])) out->angle_fam[a] = ex_a[a];
    out->best_heading = best;
    out->best_view = best_view;
    out->best_fam = ex_a[best];
    out->resolved = 1;
    return DV_OK;
}


// Host arithmetic of the key exchange; mirrors navsim_amd/sharded.py:merge_keys (the CPU tests compare the two).
extern "C" int dv_merge_keys(const uint64_t* keys, int world, int A, double delta, int signed_order, dv_merge_out* out) {
    if (!keys || !out || world < 1 || world > 64 || A < 1 || A > dv_host::kMaxHeadings) return DV_ERR_INVALID;
    memset(out, 0, sizeof(*out));
    const uint64_t top = signed_order ? 0x8000000000000000ull : 0ull;
    auto K = [&](int i) { return keys[i] ^ top; };
    auto to_double = [](uint64_t k) {
        const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
        double d;
        memcpy(&d, &b, sizeof d);
        return d;
    };
    double approx[64];
    double gmax = 0.0;
    int winner = -1;
    for (int r = 0; r < world; ++r) {
        const uint64_t w1 = K(A + dv_host::kKeyWordsPerRank * r + 1);
        if (!(w1 >> 48)) return DV_ERR_STATE;                               // that rank contributed no slot
        if (((w1 >> 32) & 0xff) >= 4) return DV_ERR_INDEX;               // that rank sensed past the end of the landscape
        approx[r] = to_double(K(A + dv_host::kKeyWordsPerRank * r));
        if (winner < 0 || approx[r] > gmax) { gmax = approx[r]; winner = r; }
    }
    long long total = 0;
    for (int r = 0; r < world; ++r) {
        if (approx[r] >= gmax - delta) {
            out->contending_mask |= 1ull << r;
            out->n_contending++;
            total += (long long)(K(A + dv_host::kKeyWordsPerRank * r + 1) & 0xffffffffull);
        }
    }
    for (int a = 0; a < A; ++a) out->angle_fam[a] = to_double(K(a));
    if (total > 1) { out->needs_resolve = 1; return DV_OK; }            // near-ties: the full records decide
    const int best = (int)K(A + dv_host::kKeyWordsPerRank * winner + 2) - 1;
    if (best < 0 || best >= A) return DV_ERR_STATE;
    out->best_heading = best;
    out->best_view = (int64_t)K(A + dv_host::kKeyWordsPerRank * winner + 3) - 1;
    out->best_fam = out->angle_fam[best];
    return DV_OK;
}


// Thermometer planes of one byte plane from the 256-bit presence map of its values: one plane per gap between
// consecutive levels, gaps wider than 127 split so that every coefficient w - 2*alpha fits an int8.
// Returns the number of planes (0 for a single level), or -1 when there are more than `cap`.
static int plan_byte_plane(const uint32_t presence[8], int cap, uint8_t* lo, uint8_t* w, int* lmin, int* lmax) {
    int levels[256], n = 0;
    for (int v = 0; v < 256; ++v)
        if (presence[v >> 5] & (1u << (v & 31))) levels[n++] = v;
    if (n == 0) { levels[n++] = 0; }                    // nothing stored (cannot happen with F >= 1): one level, 0
    *lmin = levels[0];
    *lmax = levels[n - 1];
    int t = 0;
    for (int i = 0; i + 1 < n; ++i) {
        int a = levels[i];
        const int b = levels[i + 1];
        while (a < b) {
            const int step = (b - a) > 127 ? 127 : (b - a);
            if (t >= cap) return -1;
            lo[t] = (uint8_t)a;
            w[t] = (uint8_t)step;
            ++t;
            a += step;
        }
    }
    return t;
}

// fp4 form of the matrix-core kernel: which planes stand for something and with which width.
//   planes t = 0 .. n-1 of ONE segment (HS or V), in K-element order; presence[t]: 256-bit map of the byte plane that plane
//   t belongs to; lo/w as dv_bitplane_plan left them.  A plane that starts at a library level is the first of its gap and
//   stands for the WHOLE distance to the next level (wfull); a plane that starts between levels is a copy the int8 form
//   needed (gap wider than 127: same bits) and stands for nothing (wfull = 0).  K-element n = plane n % T sits on bit n % 4
//   of a nibble: the planes on one bit position that stand for something must share their width (wacc[bit], 0 if none).
// Returns 1 if the segment qualifies, 0 if widths differ on a bit position.
static int plan_fp4_segment(int T, const uint32_t* const* presence, const uint8_t* lo, const uint8_t* w, uint8_t* wfull, int* wacc) {
    for (int t = 0; t < T; ++t) {
        const uint32_t* pres = presence[t];
        const int l = lo[t];
        int wf = 0;
        if ((pres[l >> 5] >> (l & 31)) & 1u) {
            int nxt = l + 1;
            while (nxt < 256 && !((pres[nxt >> 5] >> (nxt & 31)) & 1u)) ++nxt;
            wf = nxt < 256 ? nxt - l : (int)w[t];
        }
        wfull[t] = (uint8_t)wf;
    }
    int ok = 1;
    for (int bit = 0; bit < 4; ++bit) {
        int wb = 0;
        for (int k = 0; k < T; ++k) {                                    // (bit + 4k) % T runs through every plane on this bit
            const int wf = wfull[(bit + 4 * k) % T];
            if (!wf) continue;
            if (wb && wf != wb) ok = 0;
            wb = wf;
        }
        wacc[bit] = wb;
    }
    return ok;
}

// One byte plane's segment (all its planes from dv_bitplane_plan) through plan_fp4_segment: the CPU-testable face of it.
extern "C" int dv_fp4_plan(const uint32_t* presence, int n_planes, const uint8_t* lo, const uint8_t* w, uint8_t* wfull, int* wacc) {
    if (!presence || !lo || !w || !wfull || !wacc || n_planes < 0 || n_planes > 64) return DV_ERR_INVALID;
    const uint32_t* pres[64];
    for (int t = 0; t < n_planes; ++t) pres[t] = presence;
    for (int b = 0; b < 4; ++b) wacc[b] = 0;
    if (n_planes == 0) return 1;
    return plan_fp4_segment(n_planes, pres, lo, w, wfull, wacc);
}

extern "C" int dv_bitplane_plan(const uint32_t* presence, int cap, uint8_t* lo, uint8_t* w, int* lmin, int* lmax) {
    if (!presence || !lo || !w || !lmin || !lmax || cap < 0 || cap > 255) return DV_ERR_INVALID;
    return plan_byte_plane(presence, cap, lo, w, lmin, lmax);
}

