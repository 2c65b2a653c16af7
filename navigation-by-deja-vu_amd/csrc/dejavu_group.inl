// One process, several devices (include/dejavu.h: dv_group_*).  SURVEY 8-b2 words the boundary as
// dv_create(ctx**, device_ids[], n): a single caller thread that owns every GPU of the node.  The production multi-GPU form is one
// process per GPU (navsim_amd/sharded.py, RCCL on the step's stream); this is the same sharding behind ONE handle for a caller that
// is not a torchrun job -- e.g. the reference's single Python process with the plug-in on all GPUs of its node.
//
// A group is n contexts, member r on device_ids[r] (the same id may appear more than once: members are independent contexts, which
// is also how this file is tested on a one-GPU box).  The library is cut into contiguous blocks of views in member order, the
// np.array_split convention of navsim_amd.sharded.shard_bounds, so "first occurrence" is "lowest member, lowest local view"
// (navsim/NavBySceneFamiliarity.py:313-315).  A step: every member gets the patches (or senses its own copy from its resident
// landscape) and has its step ENQUEUED before any member is waited for -- the devices run side by side --, then the members'
// records are merged by the rules of the sharded step (dv_merge_records: maxima of the integer-sum scores over the members; a
// single candidate pair within delta of the global maximum decides; otherwise the contending members re-score their candidates
// exactly, dv_resolve, and the exact values decide with ties to the lowest view).  Host arithmetic above the C ABI only: nothing
// here touches a kernel.
struct dv_group {
    std::vector<dv_ctx*> m;
    std::vector<int64_t> lo;            // first view of member r's block (lo[n] = F)
    std::string err;
    int64_t F = 0;
    double delta = 0.0;
};

static int group_fail(dv_group* g, int rc, const char* what, const dv_ctx* c) {
    const char* msg = c ? dv_last_error(c) : nullptr;
    g->err = std::string(what) + ": " + (msg && *msg ? msg : "failed");
    return rc;
}

extern "C" int dv_group_create(dv_group** out, const int* device_ids, int n) {
    if (!out || !device_ids || n < 1 || n > 64) return DV_ERR_INVALID;
    *out = nullptr;
    dv_group* g = new (std::nothrow) dv_group();
    if (!g) return DV_ERR_OOM;
    for (int r = 0; r < n; ++r) {
        dv_ctx* c = nullptr;
        const int rc = dv_create(&c, device_ids[r]);
        if (rc) {
            for (dv_ctx* p : g->m) dv_destroy(p);
            delete g;
            return rc;
        }
        g->m.push_back(c);
    }
    g->lo.assign((size_t)n + 1, 0);
    *out = g;
    return DV_OK;
}

extern "C" void dv_group_destroy(dv_group* g) {
    if (!g) return;
    for (dv_ctx* p : g->m) dv_destroy(p);
    delete g;
}

extern "C" const char* dv_group_last_error(const dv_group* g) { return g ? g->err.c_str() : "dv_group is NULL"; }
extern "C" int dv_group_size(const dv_group* g) { return g ? (int)g->m.size() : 0; }

extern "C" int dv_group_member(dv_group* g, int r, dv_ctx** ctx, int64_t* first_view, int64_t* n_views) {
    if (!g || !ctx || r < 0 || r >= (int)g->m.size()) return DV_ERR_INVALID;
    *ctx = g->m[(size_t)r];
    if (first_view) *first_view = g->lo[(size_t)r];
    if (n_views) *n_views = g->lo[(size_t)r + 1] - g->lo[(size_t)r];
    return DV_OK;
}

extern "C" int dv_group_set_library(dv_group* g, const uint8_t* views, int64_t F, int h, int w, int channels, double cw) {
    if (!g || !views) return DV_ERR_INVALID;
    const int64_t n = (int64_t)g->m.size();
    if (F < n) { g->err = "dv_group_set_library: fewer views than members"; return DV_ERR_INVALID; }
    if (h < 1 || w < 1 || channels < 1) { g->err = "dv_group_set_library: bad shape"; return DV_ERR_INVALID; }
    const int64_t base = F / n, extra = F % n;
    for (int64_t r = 0; r <= n; ++r) g->lo[(size_t)r] = r * base + (r < extra ? r : extra);
    const size_t view_bytes = (size_t)h * w * channels;
    for (int64_t r = 0; r < n; ++r) {
        const int rc = dv_set_library(g->m[(size_t)r], views + (size_t)g->lo[(size_t)r] * view_bytes, g->lo[(size_t)r + 1] - g->lo[(size_t)r], h, w,
                                      channels, cw, g->lo[(size_t)r]);
        if (rc) return group_fail(g, rc, "dv_group_set_library", g->m[(size_t)r]);
    }
    dv_lib_info info;
    int rc = dv_get_library_info(g->m[0], &info);
    if (rc) return group_fail(g, rc, "dv_group_set_library", g->m[0]);
    g->delta = info.delta;
    g->F = F;
    return DV_OK;
}

extern "C" int dv_group_set_landscape(dv_group* g, const uint8_t* land, int rows, int cols, int channels) {
    if (!g) return DV_ERR_INVALID;
    for (dv_ctx* c : g->m) {
        const int rc = dv_set_landscape(c, land, rows, cols, channels);
        if (rc) return group_fail(g, rc, "dv_group_set_landscape", c);
    }
    return DV_OK;
}

extern "C" int dv_group_configure_sensor(dv_group* g, int sw, int sh, int pw, int ph, const uint8_t* level_lut, int mask_middle_n) {
    if (!g) return DV_ERR_INVALID;
    for (dv_ctx* c : g->m) {
        const int rc = dv_configure_sensor(c, sw, sh, pw, ph, level_lut, mask_middle_n);
        if (rc) return group_fail(g, rc, "dv_group_configure_sensor", c);
    }
    return DV_OK;
}

// func(scene, fambuf) of the plug-in (navsim/util.pyx:14-20) over the members' blocks: fambuf[F] in library order.
extern "C" int dv_group_score(dv_group* g, const uint8_t* patch, double* fambuf) {
    if (!g || !patch || !fambuf) return DV_ERR_INVALID;
    if (g->F < 1) { g->err = "dv_group_score: no library set"; return DV_ERR_STATE; }
    for (size_t r = 0; r < g->m.size(); ++r) {
        const int rc = dv_score(g->m[r], patch, fambuf + g->lo[r]);
        if (rc) return group_fail(g, rc, "dv_group_score", g->m[r]);
    }
    return DV_OK;
}

// The members' steps are enqueued (by `launch`, per member) -- all of them before the first wait --, then waited for and merged.
template <class Launch>
static int group_step(dv_group* g, int A, uint32_t flags, dv_step_result* result, double* scene_fam, Launch launch) {
    if (!g || !result) return DV_ERR_INVALID;
    if (g->F < 1) { g->err = "dv_group step: no library set"; return DV_ERR_STATE; }
    if (A < 1 || A > DV_MAX_HEADINGS) { g->err = "dv_group step: 1..64 headings"; return DV_ERR_INVALID; }
    const int n = (int)g->m.size();
    const uint32_t eflags = (flags & DV_STEP_FORCE_RESOLVE) | (scene_fam ? DV_STEP_WANT_SCENE : 0u);
    for (int r = 0; r < n; ++r) {
        int rc = launch(g->m[(size_t)r]);
        if (!rc) rc = dv_step_enqueue(g->m[(size_t)r], eflags);
        if (rc) return group_fail(g, rc, "dv_group step (enqueue)", g->m[(size_t)r]);
    }
    std::vector<dv_step_result> res((size_t)n);
    int first_rc = DV_OK;
    for (int r = 0; r < n; ++r) {                                       // every member is waited for, whatever an earlier one said
        const int rc = dv_step_wait(g->m[(size_t)r], &res[(size_t)r], scene_fam ? scene_fam + g->lo[(size_t)r] : nullptr);
        if (rc && !first_rc) first_rc = group_fail(g, rc, "dv_group step (wait)", g->m[(size_t)r]);
    }
    if (first_rc) return first_rc;
    const int64_t stride = 3 + 4 * (int64_t)A;
    std::vector<double> rec((size_t)n * (size_t)stride);
    auto pack = [&](int r) {
        const dv_step_result& s = res[(size_t)r];
        double* p = &rec[(size_t)r * (size_t)stride];
        p[0] = s.approx_max;
        p[1] = (double)s.n_candidates;
        p[2] = (s.flags & DV_RES_EXACT_ALL) ? 2.0 : ((s.flags & DV_RES_RESOLVED) ? 1.0 : 0.0);
        for (int a = 0; a < A; ++a) {
            p[3 + a] = s.angle_fam[a];
            p[3 + A + a] = (double)s.angle_view[a];
            p[3 + 2 * A + a] = s.exact_fam[a];
            p[3 + 3 * A + a] = (double)s.exact_view[a];
        }
    };
    for (int r = 0; r < n; ++r) pack(r);
    dv_merge_out out;
    int rc = dv_merge_records(rec.data(), n, A, stride, g->delta, &out);
    if (rc) { g->err = "dv_group step: dv_merge_records failed"; return rc; }
    if (out.needs_resolve) {
        // near-ties across members: the contending members that have only integer-sum scores re-score their candidates exactly
        for (int r = 0; r < n; ++r) {
            if (!((out.contending_mask >> r) & 1ull) || (res[(size_t)r].flags & (DV_RES_RESOLVED | DV_RES_EXACT_ALL))) continue;
            rc = dv_resolve(g->m[(size_t)r], &res[(size_t)r]);
            if (rc) return group_fail(g, rc, "dv_group step (resolve)", g->m[(size_t)r]);
            pack(r);
        }
        rc = dv_merge_records(rec.data(), n, A, stride, g->delta, &out);
        if (rc || out.needs_resolve) { g->err = "dv_group step: the members' exact values did not decide"; return rc ? rc : DV_ERR_STATE; }
    }
    memset(result, 0, sizeof *result);
    result->best_heading = out.best_heading;
    result->best_view = out.best_view;
    result->best_fam = out.best_fam;
    result->flags = out.resolved ? DV_RES_RESOLVED : 0u;
    result->delta = g->delta;
    result->n_headings = A;
    double gmax = res[0].approx_max;
    for (int r = 0; r < n; ++r) {
        gmax = res[(size_t)r].approx_max > gmax ? res[(size_t)r].approx_max : gmax;
        if ((out.contending_mask >> r) & 1ull) result->n_candidates += res[(size_t)r].n_candidates;
    }
    result->approx_max = gmax;
    for (int a = 0; a < DV_MAX_HEADINGS; ++a) {
        result->angle_fam[a] = a < A ? out.angle_fam[a] : 0.0;
        result->angle_view[a] = -1; result->exact_fam[a] = 0.0; result->exact_view[a] = -1;
    }
    for (int a = 0; a < A; ++a)                                         // the view of each heading's maximum: first member holding it
        for (int r = 0; r < n; ++r)
            if (res[(size_t)r].angle_fam[a] == out.angle_fam[a]) { result->angle_view[a] = res[(size_t)r].angle_view[a]; break; }
    return DV_OK;
}

extern "C" int dv_group_step(dv_group* g, const uint8_t* patches, int A, uint32_t flags, dv_step_result* result, double* scene_fam) {
    if (!patches) return DV_ERR_INVALID;
    return group_step(g, A, flags, result, scene_fam, [&](dv_ctx* c) { return dv_upload_patches(c, patches, A); });
}

extern "C" int dv_group_sense_step(dv_group* g, double x, double y, const double* angles, int A, uint32_t flags, dv_step_result* result,
                                   double* scene_fam) {
    if (!angles) return DV_ERR_INVALID;
    return group_step(g, A, flags, result, scene_fam, [&](dv_ctx* c) { return dv_sense_patches(c, x, y, angles, A); });
}
