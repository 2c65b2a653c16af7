/*
 * dejavu.h -- C ABI of the MI355X scene-familiarity engine (libdejavu_hip.so).
 *
 * This is the drop-in boundary for the one hot path of navsim
 * (Linux-cpp-lisp/navigation-by-deja-vu): scoring sensor patches against the
 * stored-view library and picking the most familiar heading.  Plain C types
 * only; every host pointer is caller-owned and borrowed for the duration of the
 * call; outputs are caller-allocated; functions return DV_OK or a negative code
 * and never throw.  A context is single-threaded (one call in flight) and bound
 * to one GPU: multi-GPU runs use one process and one context per GPU.
 *
 * Reference interfaces replaced (citations into the reference repository):
 *   navsim/util.pyx:10-25        sads_familiarity(chem_weight)(scenes) -> func
 *   navsim/util.pyx:28-73        sads_hsv_metric(familiar_scenes, scene, fambuf, chem_weight)
 *   navsim/NavBySceneFamiliarity.py:283-316   heading loop of step_forward
 *   navsim/NavBySceneFamiliarity.py:140       library hand-off at train_from_path
 */
#ifndef DEJAVU_H
#define DEJAVU_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dv_ctx dv_ctx;

enum {
    DV_OK = 0,
    DV_ERR_INVALID = -1,   /* bad argument (NULL, shape, chem_weight outside [0,1], A too large) */
    DV_ERR_HIP = -2,       /* a HIP runtime call failed; see dv_last_error */
    DV_ERR_STATE = -3,     /* no library set / no step to read back */
    DV_ERR_OOM = -4,       /* device or host allocation failed */
    DV_ERR_INDEX = -5      /* sensor footprint past the end of the landscape (the reference raises IndexError) */
};

#define DV_MAX_HEADINGS 64      /* headings scored per library pass */
#define DV_MAX_HUE_PLANES 4     /* distinct chemical hues handled by the one-hot layout */

/* dv_step flags */
#define DV_STEP_FORCE_RESOLVE 1u   /* run the exact tie resolver even for a single candidate */
#define DV_STEP_WANT_SCENE    2u   /* dv_step_enqueue: also produce scene_fam (dv_step: pass a buffer) */

/* dv_step_result.flags */
#define DV_RES_RESOLVED   1u       /* exact tie resolver ran; best_fam / exact_fam are bit-exact */
#define DV_RES_EXACT_ALL  2u       /* every score of this step is the exact sequential double */
#define DV_RES_OVERFLOW   4u       /* candidate list overflowed; step was redone in exact mode */
#define DV_RES_SENSE_ERROR 16u     /* batched calls: THIS agent's sensor footprint reached past the end of the landscape (the
                                      reference's IndexError for that trial); its other fields are meaningless.  The
                                      single-agent calls return DV_ERR_INDEX instead. */

/*
 * Result of one step (replaces NavBySceneFamiliarity.py:313-316).
 * angle_fam[a]  = max over the (local) library of the familiarity of heading a
 *                 (:313); within 1e-12 relative of the reference's double, and
 *                 bit-exact for headings whose exact_fam is finite.
 * angle_view[a] = first view index (global = first_view + local) attaining it.
 * best_heading  = np.argmax(angle_familiarity) of the reference, bit-identical
 *                 (:315): near-ties are re-scored with the reference's exact
 *                 sequential double arithmetic before deciding.
 * exact_fam[a]  = exact max over the candidate views of heading a, -inf when
 *                 heading a has no candidate (only filled when RESOLVED).
 */
typedef struct dv_step_result {
    int32_t best_heading;
    uint32_t flags;
    int64_t best_view;
    double best_fam;
    double approx_max;          /* max over headings and views of the integer-sum score */
    double delta;               /* half-width of the candidate window around approx_max */
    int64_t n_candidates;       /* (heading, view) pairs inside the window */
    int32_t n_headings;
    int32_t reserved;
    double angle_fam[DV_MAX_HEADINGS];
    int64_t angle_view[DV_MAX_HEADINGS];
    double exact_fam[DV_MAX_HEADINGS];
    int64_t exact_view[DV_MAX_HEADINGS];
} dv_step_result;

typedef struct dv_lib_info {
    int64_t n_views;            /* F (local shard) */
    int64_t first_view;         /* global index of local view 0 */
    int32_t h, w;
    int32_t n_planes;           /* bytes stored per pixel on the device */
    int32_t n_hue_planes;       /* saturation planes stored (0 when chem_weight == 0) */
    int32_t generic_hue;        /* 1: more than DV_MAX_HUE_PLANES hues, H/S kept as planes */
    int32_t has_value_plane;    /* 0 when chem_weight == 1 */
    int64_t tile_bytes;         /* bytes the scoring kernel streams per pass */
    double chem_weight;
    double delta;               /* half-width of the candidate window around the best integer-sum score */
    uint8_t hues[DV_MAX_HUE_PLANES];
    int32_t n_hues;             /* distinct hues with S > 0 in the library (entries of hues[] that are valid) */
    int32_t signed_saturation;  /* 1: two hues, S <= 127: one plane holds 128 + S(hue0) - S(hue1) */
    /* bit-plane copy for the int8 matrix-core scoring path (libraries whose stored bytes come from few levels) */
    int32_t bit_planes_hs;      /* thermometer planes of the saturation bytes (0: no bit-plane copy, or none needed) */
    int32_t bit_planes_v;       /* thermometer planes of the value byte */
    int32_t has_bit_planes;     /* 1: the bit-plane copy exists (scoring shape 6 is available) */
    int32_t fp4_form;           /* 1: the planes also allow the fp4 form of that kernel (on-level patches, see dv_patches_on_level) */
    int64_t bit_tile_bytes;     /* bytes the matrix-core scoring kernel streams per pass (int8 form; fp4 form without code tiles) */
    int64_t code_tile_bytes;    /* bytes its fp4 form streams per pass when the value plane is stored as 3-bit level codes, else 0 */
    int32_t mixed_layout;       /* 1: the saturation planes take too many values for bit planes: the matrix-core kernel scores the
                                   value bit planes, a v_sad_u8 pass the saturation BYTE planes (bit_planes_hs is 0 and a step
                                   streams bit_tile_bytes + the saturation planes' share of tile_bytes) */
    int32_t reserved0;
} dv_lib_info;

/* ---- lifetime ---------------------------------------------------------- */
int dv_create(dv_ctx **out, int device_id);
void dv_destroy(dv_ctx *ctx);
/* Message of the last failure on ctx (or of the last failed dv_create when ctx is NULL). */
const char *dv_last_error(const dv_ctx *ctx);
/* Run on a caller-provided hipStream_t.  NULL restores the context's own stream, which is NON-BLOCKING: work on the
 * legacy null stream is not ordered with it, so pass a real stream to order this context with other work. */
int dv_set_stream(dv_ctx *ctx, void *hip_stream);
/* exact != 0: every score is the reference's sequential-double value (slower fp64 kernel). */
int dv_set_exact(dv_ctx *ctx, int exact);

/* ---- library (NavBySceneFamiliarity.py:122,140; util.pyx:10-13) ------- */
/*
 * views: uint8[F,h,w,3] C-contiguous, channels H,S,V.  Copied, re-tiled into
 * the device layout and released before return.  chem_weight must be in [0,1]
 * (util.pyx:12).  first_view is the global index of views[0] when the library
 * is sharded across GPUs (0 otherwise).
 */
int dv_set_library(dv_ctx *ctx, const uint8_t *views, int64_t n_views, int h, int w,
                   int channels, double chem_weight, int64_t first_view);
/*
 * Append n more views (uint8[n,h,w,3], same shape and chem_weight) to the resident library: a further training path
 * (scripts/run_experiment.py:23-26 anticipates several per library).  Existing views keep their device tiles, only the
 * new view groups are re-tiled; kernel forms timed on the library stay chosen.  The device layout was fixed by the
 * first ingest's hue set and saturation range: views outside it are refused with DV_ERR_STATE (re-ingest everything).
 */
int dv_append_library(dv_ctx *ctx, const uint8_t *views, int64_t n_views, int channels);
/* Same library as navsim_amd.synth.synth_views(seed, n_views, h, w, first_view), generated on the GPU. */
int dv_generate_library(dv_ctx *ctx, uint64_t seed, int64_t n_views, int h, int w,
                        double chem_weight, int64_t first_view);
/* The same with the saturation drawn from every value 0..127 (a swept concentration range, scripts/run_experiment.py:132,192)
 * when full_range_s != 0: such a library keeps its saturation planes as bytes (dv_lib_info.mixed_layout).  dv_generate_patches
 * then draws its patches' saturation the same way. */
int dv_generate_library_ex(dv_ctx *ctx, uint64_t seed, int64_t n_views, int h, int w, double chem_weight, int64_t first_view,
                           int full_range_s);
int dv_clear_library(dv_ctx *ctx);
int dv_get_library_info(const dv_ctx *ctx, dv_lib_info *out);
/*
 * Host arithmetic of the bit-plane layout (no context, no GPU): the thermometer planes of one stored byte plane from
 * the 256-bit presence map of its values (bit v of presence[v / 32] = value v occurs).  One plane per gap between
 * consecutive levels, gaps wider than 127 split, so that |a - b| = const(a) + sum_t bit_t(b) * (w[t] - 2 * clamp(a -
 * lo[t], 0, w[t])) with int8 coefficients for every byte a.  Returns the number of planes written to lo[] / w[]
 * (0 for a single level), -1 when more than `cap` would be needed.
 */
int dv_bitplane_plan(const uint32_t *presence, int cap, uint8_t *lo, uint8_t *w, int *lmin, int *lmax);
/*
 * Host arithmetic of the fp4 form of the matrix-core kernel for the n_planes planes dv_bitplane_plan gave one byte plane
 * (presence: the same 256-bit map): wfull[t] = the whole gap a plane stands for when it starts at a library level (its
 * int8 copies inside a gap wider than 127 get 0), wacc[bit] = the one width of the planes that land on bit `bit` of a
 * nibble (K-element n = plane n % n_planes sits on bit n % 4).  Returns 1 when every bit position has one width (the
 * exact fp4 coefficients +-1 exist for on-level patches), 0 when not, negative on bad arguments.
 */
int dv_fp4_plan(const uint32_t *presence, int n_planes, const uint8_t *lo, const uint8_t *w, uint8_t *wfull, int *wacc);
/* Copy the stored planes of local views [v0, v0+n) back as uint8[n, n_planes, h*w] (layout check). */
int dv_read_planes(dv_ctx *ctx, int64_t v0, int64_t n, uint8_t *out);

/* ---- sensor model (NavBySceneFamiliarity.py:151-192, util.pyx:91-168) -- */
/* landscape: uint8[rows, cols, 3] HSV, C-contiguous; copied to the GPU. */
int dv_set_landscape(dv_ctx *ctx, const uint8_t *landscape, int rows, int cols, int channels);
/*
 * sensor_dimensions = [sensor_w, sensor_h], sensor_pixel_dimensions = [pixel_w, pixel_h] (:90-96);
 * lut: uint8[3][256], the per-channel level quantisation of :176-186 tabulated by the caller;
 * mask_middle_n: columns zeroed either side of the middle (:189-190).
 */
int dv_configure_sensor(dv_ctx *ctx, int sensor_w, int sensor_h, int pixel_w, int pixel_h,
                        const uint8_t *lut, int mask_middle_n);
/* get_sensor_mat at n poses; out: uint8[n, sensor_h, sensor_w, 3].  DV_ERR_INDEX like the reference's IndexError. */
int dv_sense(dv_ctx *ctx, const double *x, const double *y, const double *angle, int n, uint8_t *out);
/* The A heading patches of one position straight into the resident patches (then dv_step_enqueue).  No host
 * synchronisation: a footprint past the end of the landscape is reported by the following dv_step_wait (DV_ERR_INDEX). */
int dv_sense_patches(dv_ctx *ctx, double x, double y, const double *angles, int n_headings);
/* One agent step's device work in one call: dv_sense_patches + dv_step_enqueue + dv_step_wait. */
int dv_sense_step(dv_ctx *ctx, double x, double y, const double *angles, int n_headings, uint32_t flags,
                  dv_step_result *result, double *scene_fam);
/*
 * Ensemble form: n_agents agents (positions x[i], y[i]; headings angles[i][0..A)) sensed and scored against the one
 * resident library, 64/A agents per library pass -- the trial farm of scripts/run_experiment.py:326-347 as agents
 * batched on one GPU.  results[n_agents].  An agent whose footprint leaves the landscape gets DV_RES_SENSE_ERROR in
 * its result's flags; the other agents' results are unaffected (the reference's trials are independent).
 */
int dv_sense_step_batch(dv_ctx *ctx, const double *x, const double *y, const double *angles, int n_agents, int n_headings,
                        uint32_t flags, dv_step_result *results);
/*
 * The device work AND the device-side book-keeping of one agent step in ONE call (the fast path of
 * navsim_amd.NavBySceneFamiliarity.step_forward; a 60 us step does not want three trips through a binding):
 *   1. when an error-metric answer is outstanding (dv_path_error_enqueue of an earlier step), it is collected:
 *      *have_nearest = 1, *nearest = the distance (update_error, NavBySceneFamiliarity.py:252-276); else *have_nearest = 0;
 *   2. when do_error != 0, the metrics of position (ex, ey) -- where the PREVIOUS step ended -- are asked for
 *      (dv_path_error_enqueue: on the context's second stream, beside this step's kernels);
 *   3. the headings (angle + offsets[a]) mod 2 pi (numpy's mod: the result takes the divisor's sign, :291) are sensed at
 *      (x, y) and scored as dv_sense_step does; angle_fam[n_headings] receives the per-heading maxima, *best_heading the
 *      decision (:313-315).  DV_ERR_INDEX like the reference's IndexError.
 */
int dv_agent_step(dv_ctx *ctx, double x, double y, double angle, const double *offsets, int n_headings,
                  int do_error, double ex, double ey, double reach,
                  double *angle_fam, int32_t *best_heading, double *nearest, int32_t *have_nearest);
/*
 * dv_agent_step in two halves: _begin does 1. and 2. and LAUNCHES 3. (returns at once), _end waits for the record and hands out
 * angle_fam / best_heading (and DV_ERR_INDEX).  The heading update between two steps (NavBySceneFamiliarity.py:317-323) serialises
 * them, but everything else the host does after a step -- book-keeping, stop tests, the caller's loop -- need not wait: an agent
 * begins its next step as soon as the new pose is known.  A begun step that is never ended is harmless (its record is not read);
 * any other step call in between makes _end fail with DV_ERR_STATE.
 */
int dv_agent_step_begin(dv_ctx *ctx, double x, double y, double angle, const double *offsets, int n_headings,
                        int do_error, double ex, double ey, double reach, double *nearest, int32_t *have_nearest);
int dv_agent_step_end(dv_ctx *ctx, double *angle_fam, int32_t *best_heading);
/*
 * _end and the next _begin in one call.  The caller has worked out, while the device was busy, where every candidate heading would
 * take the agent (cand_angle[a] = (angle + offsets[a]) mod 2 pi, cand_x / cand_y[a] = position + step_size (cos, sin), :317-321); the
 * call waits for the record and begins the next step at candidate *best_heading without returning in between -- unless that
 * position fails the reference's bounds test (:153-158; bounds = {r, cols - r, rows - r}), where the next step would stop before it
 * senses: *begun = 0.  do_error != 0 asks for the error metrics of the new position with the step begun (reach as in
 * dv_agent_step); nearest / have_nearest as in dv_agent_step_begin.
 */
int dv_agent_step_end_begin(dv_ctx *ctx, double *angle_fam, int32_t *best_heading, const double *cand_x, const double *cand_y,
                            const double *cand_angle, const double *offsets, int n_headings, const double *bounds, int do_error,
                            double reach, int32_t *begun, double *nearest, int32_t *have_nearest);
/* train_from_path (:118-140) on the device: sense n poses and ingest them as the library; out_views
 * (uint8[n, sensor_h, sensor_w, 3], may be NULL) receives familiar_scenes. */
int dv_set_library_from_poses(dv_ctx *ctx, const double *x, const double *y, const double *angle, int64_t n,
                              double chem_weight, int64_t first_view, uint8_t *out_views);

/* dv_append_library with the views sensed on the device at n more poses (out_views may be NULL). */
int dv_append_library_from_poses(dv_ctx *ctx, const double *x, const double *y, const double *angle, int64_t n,
                                 uint8_t *out_views);

/* ---- error / coverage metrics (NavBySceneFamiliarity.py:252-276) -------- */
/*
 * update_error of the reference on the device.  dv_set_training_path copies the training points (double[n][2], x then
 * y; NULL or n < 1 detaches) and clears the coverage marks.  dv_path_error_enqueue computes, for one agent position, the
 * distance to every training point in the reference's double arithmetic, marks the points within `reach`
 * (coverage_threshold_factor * step_size, :271) as covered, and leaves the smallest distance for
 * dv_path_error_wait, which returns the answers in the order they were asked for (at most 8 outstanding).  Nothing
 * here waits for the GPU except dv_path_error_wait / dv_path_coverage, so the metrics of step t are computed while
 * step t+1 is scored.  dv_path_coverage copies the marks back (uint8[n], 0/1); dv_path_reset clears them and drops
 * outstanding answers (reset_error, :195-203).
 */
int dv_set_training_path(dv_ctx *ctx, const double *xy, int64_t n);
int dv_path_error_enqueue(dv_ctx *ctx, double x, double y, double reach);
int dv_path_error_wait(dv_ctx *ctx, double *nearest);
int dv_path_coverage(dv_ctx *ctx, uint8_t *out, int64_t n);
int dv_path_reset(dv_ctx *ctx);
/*
 * The same metrics for the agents of an ensemble (navsim_amd.NavEnsemble; the reference farms its trials over MPI ranks, each with
 * its own update_error): dv_path_slots gives every agent a coverage array of its own on the device (n_slots x n_path bytes, cleared;
 * 0 frees them), dv_path_error_batch is update_error for n agents at once -- nearest[i] = agent i's distance to the nearest training
 * point in the reference's double arithmetic, its slot's marks updated -- synchronous (one kernel of n x ceil(n_path / 1024) blocks
 * per 64 agents).  dv_path_coverage_slot / dv_path_reset_slot (slot < 0: all) read / clear one agent's marks.
 */
int dv_path_slots(dv_ctx *ctx, int n_slots);
int dv_path_error_batch(dv_ctx *ctx, const int32_t *slots, const double *x, const double *y, int n, double reach, double *nearest);
int dv_path_coverage_slot(dv_ctx *ctx, int slot, uint8_t *out, int64_t n);
int dv_path_reset_slot(dv_ctx *ctx, int slot);

/* ---- scoring ----------------------------------------------------------- */
/* func(scene, fambuf) of util.pyx:14-20: fambuf[f] for one patch uint8[h,w,3] against every local view. */
int dv_score(dv_ctx *ctx, const uint8_t *patch, double *fambuf);
/*
 * One navigation step's scoring (NavBySceneFamiliarity.py:283-316) for
 * patches uint8[A,h,w,3], A <= DV_MAX_HEADINGS.
 * scene_fam: double[F] or NULL -- min over headings per view (:301-303).
 */
int dv_step(dv_ctx *ctx, const uint8_t *patches, int n_headings, uint32_t flags,
            dv_step_result *result, double *scene_fam);
/*
 * Ensemble form: n_agents agents, each with its own n_headings patches (uint8[n_agents, A, h, w, 3]), against
 * the same library; results[n_agents].  Agents are scored DV_MAX_HEADINGS / A at a time in one library pass.
 */
int dv_step_batch(dv_ctx *ctx, const uint8_t *patches, int n_agents, int n_headings, uint32_t flags,
                  dv_step_result *results);
/* Re-run the exact resolver on the candidates of the last step (sharded runs, cross-rank ties). */
int dv_resolve(dv_ctx *ctx, dv_step_result *result);

/* ---- steps of ANY number of headings ------------------------------------ */
/*
 * The reference takes any n_test_angles (NavBySceneFamiliarity.py:62,87-88; its loop :289 runs over all of them, its default
 * is 60); dv_step / dv_sense_step hold at most DV_MAX_HEADINGS because their result record is a fixed structure.  The wide
 * forms take 1 <= n_headings <= DV_MAX_WIDE_HEADINGS: the headings are scored DV_MAX_HEADINGS per library pass (the passes
 * enqueued back to back, their records collected afterwards) and the passes' decisions merged on the host by the rule of
 * NavBySceneFamiliarity.py:313-315 -- the first heading attaining the maximum; where two passes' maxima lie within the
 * candidate window of each other the passes concerned are re-scored by the exact resolver first and the exact values
 * compared, so best_heading is bit-identical to np.argmax of the reference for any n_headings.
 * angle_fam[n_headings] (caller-allocated) receives every heading's maximum, angle_view[n_headings] (may be NULL) its first
 * view, scene_fam[F] (may be NULL) the minimum over ALL headings per view (:301-303).
 */
#define DV_MAX_WIDE_HEADINGS 4096
typedef struct dv_wide_result {
    int32_t best_heading;       /* index into the n_headings headings */
    uint32_t flags;             /* DV_RES_* of the pass that won */
    int64_t best_view;
    double best_fam;
    int32_t n_headings;
    int32_t n_passes;           /* library passes this step took (re-scored passes not counted) */
    int32_t n_contending;       /* passes whose maximum lay within the candidate window of the overall one */
    int32_t reserved;
} dv_wide_result;
int dv_step_wide(dv_ctx *ctx, const uint8_t *patches, int n_headings, uint32_t flags, dv_wide_result *result,
                 double *angle_fam, int64_t *angle_view, double *scene_fam);
int dv_sense_step_wide(dv_ctx *ctx, double x, double y, const double *angles, int n_headings, uint32_t flags,
                       dv_wide_result *result, double *angle_fam, int64_t *angle_view, double *scene_fam);

/* ---- ssd_f32 metric (the reference's `ssds`, navsim/util.pyx:171-184) ---- */
/*
 * views: float32[F,h,w] single channel.  Scores are sums of squared differences (smaller = more familiar):
 * fp32 fma over 16 pixels folded into a double, within ~1e-6 relative of ssds() on the upcast data; the chosen
 * heading / view are those of the exact double sums (near-ties within 3e-6 relative are re-scored exactly).
 * Up to DV_MAX_HEADINGS headings per step, scored 16 per pass over the library.  In the result, angle_fam[a] = min over views of the SSD of heading a,
 * best_heading = first heading attaining the overall minimum; scene_ssd[f] = max over headings.
 */
int dv_set_library_f32(dv_ctx *ctx, const float *views, int64_t n_views, int h, int w, int64_t first_view);
/* Same library as navsim_amd.synth.synth_views_f32(seed, n_views, h, w, first_view), generated on the GPU (BASELINE.json
 * configs[2] in its literal form is 500 000 x 128x128 float32 = 32.8 GB: made where it is scored, not uploaded). */
int dv_generate_library_f32(dv_ctx *ctx, uint64_t seed, int64_t n_views, int h, int w, int64_t first_view);
int dv_score_f32(dv_ctx *ctx, const float *patch, double *ssdbuf);
int dv_step_f32(dv_ctx *ctx, const float *patches, int n_headings, uint32_t flags, dv_step_result *result,
                double *scene_ssd);

/* ---- ssd_u8 metric (the same `ssds`, navsim/util.pyx:171-184, for uint8 views) ---- */
/*
 * views: uint8[F,h,w] single channel, h * w <= 131071.  Scores are the EXACT integer sums of squared differences -- what ssds()
 * returns for the same data as float64 -- computed on the int8 matrix cores (sum (a-b)^2 = sum a'^2 + sum b'^2 - 2 sum a'b' with
 * a' = a - 128).  Ties go to the first heading, then the first view; nothing is re-scored.  Up to DV_MAX_HEADINGS headings per step,
 * 32 per pass over the library (one byte per pixel).  Results as dv_step_f32's.
 */
int dv_set_library_u8(dv_ctx *ctx, const uint8_t *views, int64_t n_views, int h, int w, int64_t first_view);
int dv_score_u8(dv_ctx *ctx, const uint8_t *patch, double *ssdbuf);
int dv_step_u8(dv_ctx *ctx, const uint8_t *patches, int n_headings, uint32_t flags, dv_step_result *result,
               double *scene_ssd);
/*
 * The ssd_u8 metric behind the sensor model: what the agent's step needs when SSD is its familiarity plug-in
 * (navsim_amd.util.ssd_familiarity).  `channel` (0 H, 1 S, 2 V) picks the byte of each sensed HSV pixel that is compared.
 * dv_set_library_u8_from_poses = train_from_path (NavBySceneFamiliarity.py:118-140) on the device: senses n poses, hands back
 * familiar_scenes (out_views: uint8[n, sensor_h, sensor_w, 3], may be NULL) and ingests the chosen channel as the ssd_u8 library.
 * dv_sense_step_u8 = dv_sense_step for that library: the A heading patches are sensed at (x, y), their channel scored on the
 * int8 matrix cores, the least-SSD heading decided on the device; DV_ERR_INDEX like the reference's IndexError.
 */
int dv_set_library_u8_from_poses(dv_ctx *ctx, const double *x, const double *y, const double *angle, int64_t n, int channel,
                                 int64_t first_view, uint8_t *out_views);
int dv_sense_step_u8(dv_ctx *ctx, double x, double y, const double *angles, int n_headings, int channel, uint32_t flags,
                     dv_step_result *result, double *scene_ssd);

/* ---- resident / asynchronous form (benchmarks, pipelined callers) ------ */
/* Upload patches and prepare their device layout; no host synchronisation. */
int dv_upload_patches(dv_ctx *ctx, const uint8_t *patches, int n_headings);
/* Generate patches on the device: synth_views(seed + 1, n_headings, h, w). */
int dv_generate_patches(dv_ctx *ctx, uint64_t seed, int n_headings);
/* Enqueue one step on the resident patches (scoring, reductions, tie resolver); no host synchronisation. */
int dv_step_enqueue(dv_ctx *ctx, uint32_t flags);
/* Wait for the last enqueued step and copy its result (and optionally scene_fam[F]). */
int dv_step_wait(dv_ctx *ctx, dv_step_result *result, double *scene_fam);
/*
 * Device address of the packed record of the last enqueued step, for a device-side exchange between ranks
 * (stream-ordered after dv_step_enqueue on the context's stream; no host synchronisation):
 *   double[3 + 4A] = approx_max, n_candidates, state (0 integer scores, 1 candidates exact, 2 all exact; + 4 when the
 *                    patches were sensed past the end of the landscape: dv_merge_records then returns DV_ERR_INDEX),
 *                    angle_fam[A], angle_view[A], exact_fam[A], exact_view[A]
 */
int dv_step_record(dv_ctx *ctx, void **device_ptr, int *n_doubles);
/*
 * The sharded step's fast exchange: packed keys of the last enqueued step for ONE all-reduce(max) of uint64 words
 * (north star: one all-reduce per navigation step).  dv_step_keys enqueues, behind the step, the kernel that builds
 *   keys[a], a < A              ordered image of this rank's max over its views of heading a's score
 *   keys[A + 4r .. A + 4r + 3]  rank r's slot (zero on the other ranks): ordered image of its best score; candidate
 *                               count | state << 32 | 1 << 48; its first-maximum heading + 1; that heading's view + 1
 * and returns their device address (n_words = A + 4 * world).  signed_order != 0 flips every word's top bit, so that a
 * signed 64-bit maximum orders them correctly.  dv_merge_keys (host arithmetic only, no context) takes the reduced
 * words: when a single (heading, view) pair lies within delta of the global maximum it fills the decision, else it
 * sets needs_resolve and the full records are exchanged (dv_step_record / dv_merge_records).  DV_ERR_INDEX when a rank
 * reports patches sensed past the end of the landscape.
 */
int dv_step_keys(dv_ctx *ctx, int rank, int world, int signed_order, void **device_ptr, int *n_words);
/* Enqueue the exact resolver on the last step's candidates (updates the record); no host synchronisation. */
int dv_resolve_enqueue(dv_ctx *ctx);
/*
 * Hand a device buffer (e.g. the output of an all-gather of the ranks' records) to the host without a blocking
 * stream synchronisation: dv_publish enqueues, on the context's stream, a copy of n_doubles doubles into mapped host
 * memory followed by a sequence word; dv_publish_wait polls that word and copies the doubles to dst.
 */
int dv_publish(dv_ctx *ctx, const void *device_src, int64_t n_doubles);
int dv_publish_wait(dv_ctx *ctx, double *dst, int64_t n_doubles);

/*
 * Mailbox exchange for the ranks of ONE node (an alternative to an all-gather: no collective kernel at all).
 * host_base: a page-aligned host segment that every rank's process has mapped (e.g. a file in /dev/shm), laid out as
 * [slots][world][512 doubles]; dv_set_mailbox registers it with this context's GPU (NULL detaches).
 * dv_mailbox_post enqueues, behind the step's kernels, the copy of this rank's packed record (dv_step_record) into
 * entry [slot][rank] with sequence number seq; dv_mailbox_wait polls (host) until the ranks of rank_mask have posted
 * seq into `slot` and copies their records to records[r * stride ..] (timeout_ms <= 0: wait for ever).  A slot may be
 * reused once every rank has read it: two slots per exchange round alternate safely, because nobody posts step
 * s + 2 before everybody has posted step s + 1, i.e. has finished reading step s.
 */
int dv_set_mailbox(dv_ctx *ctx, void *host_base, int64_t bytes, int rank, int world);
int dv_mailbox_post(dv_ctx *ctx, int slot, uint64_t seq);
int dv_mailbox_wait(dv_ctx *ctx, int slot, uint64_t seq, uint64_t rank_mask, double *records, int64_t stride, int timeout_ms);

/*
 * The global decision from the gathered per-rank records (the sharded form of NavBySceneFamiliarity.py:313-316);
 * identical on every rank.  records[r * stride .. ] is rank r's record as laid out by dv_step_record.  Host
 * arithmetic only: no context, no GPU.  When ranks that contend for the maximum (approx_max within delta of the
 * global one) hold more than one candidate pair between them and one of them has not re-scored its candidates yet,
 * needs_resolve is set and no decision is made: those ranks (contending_mask, bit r = rank r, world <= 64) run
 * dv_resolve_enqueue and the records are exchanged again.
 */
typedef struct dv_merge_out {
    int32_t best_heading;
    int32_t resolved;           /* 1: exact values decided; 0: a single candidate pair, integer scores decided */
    int64_t best_view;
    double best_fam;
    int32_t needs_resolve;
    int32_t n_contending;
    uint64_t contending_mask;
    double angle_fam[DV_MAX_HEADINGS];
} dv_merge_out;
int dv_merge_records(const double *records, int world, int n_headings, int64_t stride, double delta, dv_merge_out *out);
int dv_merge_keys(const uint64_t *keys, int world, int n_headings, double delta, int signed_order, dv_merge_out *out);
int dv_synchronize(dv_ctx *ctx);

/* ---- one process, several devices ---------------------------------------- */
/*
 * SURVEY 8-b2 words the boundary as dv_create(ctx**, device_ids[], n): one caller thread that owns all GPUs of a node (the
 * reference is a single Python process, navsim/NavBySceneFamiliarity.py:72,140,299).  A dv_group is that form: n contexts, member r
 * on device_ids[r], behind one handle.  The library is cut into contiguous blocks of views in member order (first members take
 * the extras), every member's step is enqueued before any is waited for, and the members' records are merged by the sharded
 * step's rules (dv_merge_records; near-ties across members are re-scored exactly by the contending members, dv_resolve) -- the
 * decision is the unsharded one, first occurrence included (:313-315).  The multi-PROCESS form (one rank per GPU, one RCCL
 * all-reduce per step) is navsim_amd/sharded.py over dv_step_keys / dv_step_record.
 *   dv_group_step        patches uint8[A, h, w, 3] go to every member; scene_fam (may be NULL) receives float64[F] in library order
 *   dv_group_sense_step  every member senses the patches from its own copy of the landscape (dv_group_set_landscape,
 *                        dv_group_configure_sensor): nothing but the pose goes up
 *   dv_group_score       func(scene, fambuf) of the plug-in (util.pyx:14-20): fambuf[F]
 *   result               best_heading / best_view (global index) / best_fam, angle_fam[A] and angle_view[A] (maxima over all
 *                        members and their first views), flags DV_RES_RESOLVED when exact values decided; exact_* unused
 * The same device id may be given more than once (independent contexts on one GPU).
 */
typedef struct dv_group dv_group;
int dv_group_create(dv_group **out, const int *device_ids, int n_devices);
void dv_group_destroy(dv_group *g);
const char *dv_group_last_error(const dv_group *g);
int dv_group_size(const dv_group *g);
int dv_group_member(dv_group *g, int r, dv_ctx **ctx, int64_t *first_view, int64_t *n_views);
int dv_group_set_library(dv_group *g, const uint8_t *views, int64_t n_views, int h, int w, int channels, double chem_weight);
int dv_group_set_landscape(dv_group *g, const uint8_t *landscape, int rows, int cols, int channels);
int dv_group_configure_sensor(dv_group *g, int sensor_w, int sensor_h, int pixel_w, int pixel_h, const uint8_t *level_lut,
                              int mask_middle_n);
int dv_group_score(dv_group *g, const uint8_t *patch, double *fambuf);
int dv_group_step(dv_group *g, const uint8_t *patches, int n_headings, uint32_t flags, dv_step_result *result, double *scene_fam);
int dv_group_sense_step(dv_group *g, double x, double y, const double *angles, int n_headings, uint32_t flags,
                        dv_step_result *result, double *scene_fam);

/* ---- measurement ------------------------------------------------------- */
/* hipEvent pair on the context's stream around whatever is enqueued between the two calls. */
int dv_timer_start(dv_ctx *ctx);
int dv_timer_stop(dv_ctx *ctx, float *elapsed_ms);
/* enable = n > 0: bracket every n-th launch of the scoring kernel with hipEvents (1 = every launch; an event pair
 * costs a few microseconds of stream time, so throughput runs sample); 0: off. */
int dv_profile_kernel(dv_ctx *ctx, int enable);
/* Sum and count of the bracketed scoring-kernel launches since the last read; resets. */
int dv_profile_read(dv_ctx *ctx, double *total_ms, int64_t *n_launches);
/* Workgroup shape of the scoring kernel in use for steps of n_headings headings on the resident library (1..5: forms
 * of the byte-plane kernels, csrc/dejavu_hip.hip:launch_tiles_apad; 6: the bit-plane matrix-core kernel k_sad_mfma_dual (mixed layout: + a v_sad_u8 pass over the saturation byte planes);
 * 0 = not timed yet, or not applicable to this library). */
int dv_workgroup_shape(dv_ctx *ctx, int n_headings, int *shape);
/* Streaming-read microbenchmark over n_bytes of device memory (achievable HBM ceiling). */
int dv_stream_read_gbps(dv_ctx *ctx, int64_t n_bytes, int iters, double *gbps);

/*
 * roctx ranges (rocprofv3 --marker-trace): with DEJAVU_ROCTX=1 in the environment the library brackets the phases of a
 * step ("dv:sense", "dv:score", "dv:finish", "dv:wait") and callers can add their own (navsim_amd/sharded.py wraps the
 * per-step exchange in "dv:exchange").  No-ops otherwise; libroctx64.so is looked up at run time.
 */
int dv_range_push(const char *name);
int dv_range_pop(void);

/* 1 when the resident patches have every byte on one of the library's levels (or outside their range), i.e. the last
 * coefficient prep allowed the fp4 form of the matrix-core kernel; 0 when a byte lies strictly inside a gap (the int8
 * form scored or will score them); DV_ERR_STATE when the library has no fp4 form or no patches were prepared for it.
 * Waits for the stream.  Diagnostic: the kernel itself reads the same word on the device, nothing on the host decides.
 * (The reference has no counterpart; its patches come out of the quantiser of NavBySceneFamiliarity.py:176-186.) */
int dv_patches_on_level(dv_ctx *ctx);

/* Form of the last integer scoring pass, as flags: DV_FORM_MATRIX_CORES (bit-plane library on the matrix cores, workgroup
 * shape 6), DV_FORM_FP4 (its fp4 coefficients: the prep's patches were on-level), DV_FORM_FUSED_FINISH (the kernel
 * finished its scores itself; the step ended in k_fold).  Waits for the stream when the fp4 word has to be read.
 * Diagnostic for benchmarks and tests; negative on error. */
#define DV_FORM_MATRIX_CORES 1
#define DV_FORM_FP4 2
#define DV_FORM_FUSED_FINISH 4
int dv_scoring_form(dv_ctx *ctx);

const char *dv_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DEJAVU_H */
