// dejavu_hip.hip -- host side of libdejavu_hip.so: the C ABI of include/dejavu.h.
//
// One context = one GPU = one stream.  All device buffers are allocated when the library is set (nothing is
// allocated inside a step) and results come back through one pinned, mapped host record that the host polls.  A
// single-agent step on the default path is three launches on one stream:
//   k_patch_prep (sensing / upload / generator -> raw bytes, byte operands, coefficient images, per-heading constants)
//   -> k_sad_mfma_dual (bit tiles on the matrix cores, scores finished in its epilogue: one summary per workgroup)
//   -> k_fold (folds the <= 256 summaries, decides, writes the record)
// Libraries the bit planes cannot describe, steps that want scene_fam and K-chunked small libraries end in k_finish
// (sums -> scores, reductions, decision) behind k_sad_tiles / k_sad_packed / k_sad_generic or the unfused matrix-core
// pass; dv_score, the exact mode, ssd_f32 and ssd_u8 keep scores in fam[] (scoring kernel -> k_combine* / k_exact_all
// -> k_tail).  Only when near-ties need exact re-scoring: k_resolve -> k_decide.
#include "dejavu_kernels.h"
#include "../../include/dejavu.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <limits>
#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace dv;

static_assert(sizeof(StepResultDev) == sizeof(dv_step_result) + sizeof(unsigned long long) &&
              offsetof(StepResultDev, check) == sizeof(dv_step_result), "device record = host record + check word");
static_assert(kMaxHeadings == DV_MAX_HEADINGS && kMaxHues == DV_MAX_HUE_PLANES, "header constants differ");
static_assert(kResSenseError == DV_RES_SENSE_ERROR, "header constants differ");

#include "dejavu_host.inl"      // dv_merge_records, dv_merge_keys, dv_bitplane_plan: host arithmetic, no HIP

#include <dlfcn.h>

// roctx ranges around the phases of a step (sense / score / finish / wait; the Python exchange brackets its collective
// through dv_range_push / dv_range_pop), visible in rocprofv3 --marker-trace.  libroctx64.so is looked up at run time and
// only when DEJAVU_ROCTX=1 (librocprofiler-sdk-roctx.so, else libroctx64.so): no link-time dependency on the tracer and the ranges cost one predictable
// branch otherwise.
namespace {
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char* e = getenv("DEJAVU_ROCTX");
        if (!e || atoi(e) == 0) return;
        // rocprofv3 (rocprofiler-sdk) records the ranges of its own roctx library; the older roctracer one is the fallback
        void* h = nullptr;
        for (const char* name : {"librocprofiler-sdk-roctx.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so", "libroctx64.so",
                                 "/opt/rocm/lib/libroctx64.so"}) {
            h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (h) break;
        }
        if (!h) return;
        push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) { push = nullptr; pop = nullptr; }
    }
};
Roctx& roctx() { static Roctx r; return r; }
struct Range {
    bool on;
    explicit Range(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
    ~Range() { if (on) roctx().pop(); }
};
}  // namespace

extern "C" int dv_range_push(const char* name) {
    if (!name) return DV_ERR_INVALID;
    if (roctx().push) roctx().push(name);
    return DV_OK;
}
extern "C" int dv_range_pop(void) {
    if (roctx().pop) roctx().pop();
    return DV_OK;
}

static thread_local std::string g_create_error;

struct dv_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipStream_t aux_stream = nullptr;         // the agent's error / coverage metrics (k_path_error): beside the steps, not between them
    std::string err;
    int exact = 0;

    // library
    int metric = 0;                           // 0 = sads_hsv (uint8 HSV), 1 = ssd_f32 (float32, one channel), 2 = ssd_u8 (uint8, one channel)
    float4* d_ftiles = nullptr;               // ssd_f32 library
    float* d_fraw = nullptr;                  // [64][P] raw float patches
    float* d_fprep = nullptr;                 // [Q][4][64]
    double* d_fpart = nullptr;                // [nchunk][64][Fpad]
    float4* d_fprep4 = nullptr;               // ssd_f32 on the matrix cores: [Q][APAD] float4 patch rows (k_prep_f32x)
    double* d_fvnorm = nullptr;               // [Fpad] sum of squares of each view (k_norm_f32, at ingest)
    double* d_fpnorm = nullptr;               // [64] ... of each heading's patch
    unsigned long long* d_flower = nullptr;   // [64][32 shards] per heading the best lower bound of the step (ordered key; k_combine_f32x)
    uint4* d_fprepb = nullptr;                // [2 passes][ceil(P/16)][hi, lo][64] bf16 operand rows of the two-term form (k_prep_f32b)
    double f32x_kappa = 0.0;                  // error bound of the form the last cross-term pass took (k_cand_f32x)
    bool fprep_direct_ready = false;          // d_fprep (the direct form's operand layout) describes the resident patches
    int nt_env = -1;                          // DEJAVU_NT=0/1: default / non-temporal policy for the matrix-core kernel's library rows (default: by size)
    int ssd_mfma_env = 1;                     // DEJAVU_SSD_MFMA=0: ssd_f32 steps keep the direct form (k_ssd_tiles) throughout
    bool f32x_request = false;                // enqueue_step: this ssd_f32 pass may take the cross-term form (no per-view output wanted)
    bool f32x_used = false;                   // ... and did: the step ends in k_cand_f32x + k_resolve_f32 + k_decide
    uint4* d_u8tiles = nullptr;               // ssd_u8 library: [Fpad/32][K][64] (k_retile_u8), K = ceil(P / 32) K-steps
    unsigned char* d_u8raw = nullptr;         // [64][P] raw uint8 patches
    uint4* d_u8prep = nullptr;                // [2][K][64] operand rows of the two passes of 32 headings
    int* d_u8part = nullptr;                  // [u8_nchunk][64][Fpad] cross terms
    unsigned long long* d_vnorm = nullptr;    // [Fpad] sum of (x - 128)^2 of each view
    unsigned long long* d_pnorm = nullptr;    // [64] ... of each heading's patch
    int u8_K = 0, u8_KC = 0, u8_nchunk = 0;   // K-steps, K-steps per LDS chunk, chunks
    bool have_lib = false;
    LibCfg cfg{};
    int h = 0, w = 0;
    uint4* d_tiles = nullptr;
    size_t tile_bytes = 0;

    // per-step buffers (sized at set_library)
    unsigned char* d_raw_patches = nullptr;   // [64][P][3]
    unsigned* d_prep = nullptr;               // [npl][Q][4][64]
    PrepAcc* d_acc = nullptr;                 // [2] per-heading constants, off-level word and sensor-error mask of a patch preparation:
    int acc_parity = 0;                       //   the sets alternate (k_patch_prep clears the other one); d_acc[acc_parity] is the resident patches'
    PrepBits pbits{};                         // what k_patch_prep needs of the bit planes (smallest level, on-level bitmaps)
    unsigned* d_one = nullptr;                // a word that holds 1: the "off level" word of libraries without an fp4 form
    unsigned long long* d_bsum2 = nullptr;    // k_fold_reduce: [agents][kFoldSlices][2][headings per agent]
    unsigned long long* d_bsum = nullptr;     // k_finish: per-block, per-heading (maximum, first view) [blocks][2][headings]
    unsigned long long* d_ctmp = nullptr;     // k_finish: shared extra-candidate list [agents][kTmpCap][2]
    int int_has_hs = 0, int_has_v = 0;        // which sums the last integer scoring pass produced
    int finish_vb_env = 0;                    // DEJAVU_FINISH_VB: view sets of 256 per k_finish block
    int fenced_env = -1;                      // DEJAVU_FENCED: 1 release/acquire around the arrival ticket always, 0 never; default by path
    int finish_fused = 1;                     // DEJAVU_FINISH: integer-path steps end in k_finish where it pays (0: never, 2: whenever possible)
    unsigned* d_part = nullptr;               // [nchunk][nsum][APAD][Fpad] raw integer sums of one pass
    unsigned long long* d_pmax = nullptr;     // [64][max(G, Fpad/256)] partial maxima
    int n_partial = 0;                        // partial maxima per heading left in d_pmax by the last scoring
    int nchunk = 1;                           // pixel chunks per view group of the last launch (work items = G * nchunk)
    int nchunk_cap = 1;                       // chunks the partial-sum buffer has room for
    int target_items = 0;                     // 0 = as many items as waves are resident (DEJAVU_TARGET_ITEMS overrides)
    int waves_per_cu = 0;                     // resident waves per CU the grid is sized for; 0 = by kernel (DEJAVU_WPC)
    int shape_env = 0;                        // DEJAVU_SHAPE: workgroup shape of k_sad_tiles, 1..4 (0: timed once per library)
    int force_shape = 0;                      // set while tune_workgroup_shape() times a candidate
    int tuned_shape[4] = {0, 0, 0, 0};        // chosen shape per heading class (8, 16, 32, 64 resident); 0 = not timed yet
    float tuned_us[4][6] = {};                // what the timing saw per shape (DEJAVU_VERBOSE prints it)
    // bit-plane copy of the library for the int8 MFMA scoring path (k_sad_mfma), when its values come from few levels
    bool bits_ok = false;
    bool mixed = false;                       // the bit tiles hold the V segment only; the saturation planes are scored as bytes (launch_int_scoring)
    bool hs_bytes_pass = false;               // set while that byte pass is being launched (scoring_grid: one chunk, like the matrix-core pass)
    BitCfg bcfg{};
    uint4* d_btiles = nullptr;                // [Fpad/32][GS][64]
    size_t btile_bytes = 0;                   // of which streamed per pass: (Fpad/32) * (NK_hs + NK_v) KB
    uint4* d_coef = nullptr;                  // [2 passes][NK][8][64] int8 coefficient image of the resident patches
    bool coef_ready = false;                  // d_coef / d_coef4 describe the resident patches
    bool prep_dwords_ready = false;           // d_prep describes them too (k_patch_prep leaves it out for steps on the matrix cores)
    int bits_env = 1;                         // DEJAVU_BITS: 0 never build the bit planes, 1 when they save bytes, 2 whenever possible
    int mfma_tiles_env = 0, mfma_chunk_env = 0;   // DEJAVU_MFMA_TILES / DEJAVU_MFMA_CHUNK (0 = by library size)
    const int* int_hsconst = nullptr;         // constants that go with the partial sums of the last integer scoring pass
    const int* int_vconst = nullptr;
    bool fuse_request = false;                // enqueue_step: this pass may finish its scores inside the scoring kernel
    bool epilogue_fused = false;              // the last integer scoring pass finished its scores itself (k_sad_mfma_ring<.., true>): only k_fold is left
    // fp4 form of the matrix-core kernel (sad_ring_fp4): possible when every plane of a segment has one gap width
    bool fp4_ok = false;                      // this library's planes qualify (build_bit_planes)
    int fp4_env = 1;                          // DEJAVU_FP4=0: never
    uint4* d_coef4 = nullptr;                 // [pass][K-step][4][64] E2M1 sign images (k_patch_prep)
    uint4* d_ctiles = nullptr;                // [Fpad/32][GSC][64] code tiles (k_bitpack_code), when bcfg.vcode
    size_t ctile_bytes = 0;
    // DEJAVU_VCODE=1: the fp4 form reads five-level value planes as 3-bit codes (k_bitpack_code: a third copy of the library, 5
    // bits per pixel).  Off by default: at 500 000 views x 128x128 it moves 5.16 GB instead of 6.18 GB per pass in the SAME
    // 0.94 ms -- with the library stream out of the way the loop is bound by its LDS operand traffic and the matrix pipe
    // (tools/exp/fp4_ladder.hip), so the copy would cost 5 GB and buy nothing yet.
    int vcode_env = 0;
    int last_form = 0;                        // DV_FORM_* of the last integer scoring pass (fp4 bit: the fp4 image existed)
    int fuse_env = 1;                         // DEJAVU_FUSE=0: one-chunk matrix-core passes leave their sums to k_finish instead of finishing them
    int lc_env = 1;                           // DEJAVU_LC: 0 every wave loads and multiplies, 1 loader / consumer waves (stages of 4, ring of 3), 2 (stages of 2, ring of 5)
    int ht_env = 2;                           // DEJAVU_HT=1: 64 resident headings as two passes instead of two heading tiles per view group
    int ring_env = 0;                         // DEJAVU_RING: ring shapes of the round-2 body (A/B)
    int mixed_env = 1;                        // DEJAVU_MIXED=0: no mixed layout (many saturation values keep the byte path)
    int tune_all_env = 0;                     // DEJAVU_TUNE_ALL=1: time every byte-path shape even when the matrix-core pass already beats their byte bound
    int fail_alloc_env = 0;                   // DEJAVU_TEST_FAIL_ALLOC=n: the n-th allocation of an ssd_u8 / ssd_f32 ingest fails (tests of the clean-up path)
    int lib_allocs = 0;                       // ... counted here
    int fused_nb = 0;                         // summaries per agent it left
    int group_pad_kb = -1;                    // DEJAVU_GPAD, see group_stride
    int allow_signed = 1;                     // DEJAVU_SIGNED=0 keeps two one-hot saturation planes even when one signed plane would do
    double* d_fam = nullptr;                  // [64][Fpad]
    double* d_scene = nullptr;                // [Fpad]
    StepState* d_state = nullptr;
    unsigned long long* d_cand = nullptr;     // [kCandCap]
    double* d_cand_exact = nullptr;           // [kCandCap]
    int result_slot = 0;                      // first result record of the pass being enqueued (pipelined ensemble passes)
    // Ensemble passes packed over the whole chip (run_batch): every pass of a group gets its OWN set of the per-step buffers, so that
    // all preparations run first, the scoring kernels of consecutive passes then sit in two hardware queues and fill each other's last,
    // partly empty round of work items (782 items on 256 workgroups: 14 run a fourth), and the folds come at the end.
    // `extra[j - 1]` holds set j while it is not in the fields above; set 0 is the context's own.
    struct PassSet {
        unsigned char* raw = nullptr; uint4* coef = nullptr; uint4* coef4 = nullptr; PrepAcc* acc = nullptr; int parity = 0;
        unsigned long long* bsum = nullptr; unsigned long long* ctmp = nullptr; StepState* state = nullptr;
        unsigned long long* cand = nullptr; double* cand_exact = nullptr;
        bool coef_ready = false, prep_dwords_ready = false, patches_sensed = false;
        int A = 0, APAD = 0, n_agents = 1, A_agent = 0, fused_nb = 0;
    };
    static constexpr int kExtraSets = 8;
    PassSet extra[kExtraSets];
    int n_extra = 0;                          // extra sets allocated (at the first ensemble call on this library that wants them)
    int cur_set = 0;                          // which set is in the fields
    hipStream_t batch_stream = nullptr;       // the second hardware queue of the scoring kernels
    hipStream_t batch_stream3 = nullptr;      // a third one (DEJAVU_CHAINS=3, A/B)
    hipEvent_t ev_a = nullptr, ev_b = nullptr;
    bool defer_fold = false;                  // enqueue_step: a fused pass leaves its fold to the caller (run_batch launches them last)
    int deferred_force = 0, deferred_seq = 0; // ... with these arguments
    int lc22_env = 1;                         // DEJAVU_LC22=0: passes of 64 headings keep one view group per consumer (sad_lc_fp4 with two heading tiles; A/B)
    int chain_order_env = -1;                 // DEJAVU_CHAIN_ORDER (A/B): how a chain of ensemble passes is laid out on its stream, see run_batch
    int chains_env = 2;                       // DEJAVU_CHAINS=1: ensemble passes one after the other on one stream, as in round 3 (A/B)
    StepResultDev* h_result = nullptr;        // pinned, mapped: the kernels write the result record into it
    StepResultDev* d_result = nullptr;        // device-side address of h_result
    double* d_record = nullptr;               // [3 + 4*64] packed record of the last step, for device-side exchange
    unsigned long long* d_keys = nullptr;     // [64 + 4*64] packed keys of the last step (all-reduce(max) exchange)
    double* h_scene = nullptr;                // pinned staging for scene_fam
    int A = 0, APAD = 0;                      // resident patches (all agents of the pass)
    int n_agents = 1, A_agent = 0;            // agents in the resident pass and headings per agent
    bool step_pending = false;
    bool patches_sensed = false;              // resident patches were produced by k_sense (its error flag is live)
    int seq = 0;                              // sequence number of the last enqueued pass (written back by k_tail)
    double* h_mbox = nullptr;                 // mailbox exchange: the node's shared host segment [slots][world][kMboxEntry]
    double* d_mbox = nullptr;                 // its address on this GPU (hipHostRegister)
    int mbox_rank = 0, mbox_world = 0, mbox_slots = 0;
    double* h_pub = nullptr;                  // mapped host buffer of dv_publish: [8 doubles: sequence word][payload]
    double* d_pub = nullptr;                  // its device address
    int64_t pub_cap = 0;                      // payload capacity in doubles
    unsigned long long pub_seq = 0;
    int spin_wait = 1;                        // poll the mapped result record instead of blocking on the stream (DEJAVU_SPIN)
    bool last_want_scene = false;
    double delta = 0.0;

    // sensor model (landscape resident in HBM)
    unsigned char* d_land = nullptr;
    SensorCfg sensor{};
    bool have_sensor = false;
    unsigned char* d_lut = nullptr;           // [3][256] level-quantisation tables
    Pose* d_poses = nullptr;
    size_t poses_cap = 0;
    unsigned char* d_sense = nullptr;         // [n][sh][sw][3] scratch for dv_sense
    size_t sense_cap = 0;
    unsigned long long* d_sense_err = nullptr;   // k_sense's out-of-bounds flag (dv_sense, ingest from poses)
    std::vector<Pose> h_poses;

    // error / coverage metrics of the agent (training path resident; answers collected one step later)
    double* d_path = nullptr;                 // [n][2]
    int64_t n_path = 0;
    unsigned char* d_cover = nullptr;         // [n] coverage marks
    unsigned char* d_cover_slots = nullptr;   // [n_cover_slots][n] coverage marks of an ensemble's agents (dv_path_slots)
    int n_cover_slots = 0;
    unsigned long long* d_minkeys = nullptr;  // [kPathBatch] dv_path_error_batch's minima
    unsigned long long* h_minkeys = nullptr;  // pinned
    PathErrState* d_errstate = nullptr;
    PathErrOut* h_errout = nullptr;           // mapped ring of kErrRing answers
    PathErrOut* d_errout = nullptr;
    unsigned long long err_enq = 0, err_deq = 0;   // answers requested / collected
    int agent_pending = 0;                    // headings of the agent step begun and not yet ended (dv_agent_step_begin / _end)
    bool err_on_main = false;                 // the last metric computation rode on the step's own stream (dv_agent_step)

    // measurement
    hipEvent_t t0 = nullptr, t1 = nullptr;
    int profile = 0;                          // dv_profile_kernel: bracket every profile-th scoring launch with events
    long long profile_count = 0;
    std::vector<hipEvent_t> pev;              // pairs
    size_t pev_used = 0;
};

static int fail(dv_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(ctx, e_ == hipErrorOutOfMemory ? DV_ERR_OOM : DV_ERR_HIP, "%s: %s (%s:%d)", #expr, \
                        hipGetErrorString(e_), __FILE__, __LINE__);                             \
    } while (0)

static void use_set(dv_ctx* c, int which);
static void free_library(dv_ctx* c) {
    auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
    use_set(c, 0);
    for (int j = 0; j < dv_ctx::kExtraSets; ++j) {
        dv_ctx::PassSet& a = c->extra[j];
        F(a.raw); F(a.coef); F(a.coef4); F(a.acc); F(a.bsum); F(a.ctmp); F(a.state); F(a.cand); F(a.cand_exact);
        a = dv_ctx::PassSet{};
    }
    c->n_extra = 0;
    F(c->d_tiles); F(c->d_raw_patches); F(c->d_prep); F(c->d_acc); F(c->d_one); F(c->d_fam); F(c->d_scene);
    F(c->d_part); F(c->d_pmax); F(c->d_record); F(c->d_keys); F(c->d_bsum); F(c->d_bsum2); F(c->d_ctmp);
    F(c->d_ftiles); F(c->d_fraw); F(c->d_fprep); F(c->d_fpart); F(c->d_fprep4); F(c->d_fvnorm); F(c->d_fpnorm); F(c->d_flower); F(c->d_fprepb);
    F(c->d_u8tiles); F(c->d_u8raw); F(c->d_u8prep); F(c->d_u8part); F(c->d_vnorm); F(c->d_pnorm);
    F(c->d_btiles); F(c->d_coef); F(c->d_coef4); F(c->d_ctiles); c->ctile_bytes = 0;
    c->pbits = PrepBits{};
    c->bits_ok = false; c->coef_ready = false; c->btile_bytes = 0;
    c->metric = 0;
    F(c->d_state); F(c->d_cand); F(c->d_cand_exact);
    if (c->h_result) { (void)hipHostFree(c->h_result); c->h_result = nullptr; c->d_result = nullptr; }
    if (c->h_scene) { (void)hipHostFree(c->h_scene); c->h_scene = nullptr; }
    c->have_lib = false;
    for (int i = 0; i < 4; ++i) c->tuned_shape[i] = 0;
    c->A = 0;
    c->step_pending = false;
}

extern "C" const char* dv_version(void) { return "dejavu-mi355x 0.1 (gfx950)"; }

extern "C" const char* dv_last_error(const dv_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int dv_create(dv_ctx** out, int device_id) {
    if (!out) return fail(nullptr, DV_ERR_INVALID, "dv_create: out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(nullptr, hipGetDeviceCount(&n));
    if (device_id < 0 || device_id >= n) return fail(nullptr, DV_ERR_INVALID, "dv_create: device %d of %d", device_id, n);
    HIP_TRY(nullptr, hipSetDevice(device_id));
    dv_ctx* c = new (std::nothrow) dv_ctx();
    if (!c) return fail(nullptr, DV_ERR_OOM, "dv_create: out of host memory");
    c->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&c->t0);
    if (e == hipSuccess) e = hipEventCreate(&c->t1);
    if (e != hipSuccess) {
        fail(nullptr, DV_ERR_HIP, "dv_create: %s", hipGetErrorString(e));
        delete c;
        return DV_ERR_HIP;
    }
    c->stream = c->own_stream;
    auto env_int = [](const char* name, int& dst, int lo, int hi) {
        if (const char* t = getenv(name)) { const int v = atoi(t); if (v >= lo && v <= hi) dst = v; }
    };
    env_int("DEJAVU_TARGET_ITEMS", c->target_items, 1, 1 << 24);
    env_int("DEJAVU_WPC", c->waves_per_cu, 1, 32);
    env_int("DEJAVU_SHAPE", c->shape_env, 0, 6);
    env_int("DEJAVU_BITS", c->bits_env, 0, 2);
    env_int("DEJAVU_MFMA_TILES", c->mfma_tiles_env, 0, 2);
    env_int("DEJAVU_MFMA_CHUNK", c->mfma_chunk_env, 0, 32);
    env_int("DEJAVU_FINISH", c->finish_fused, 0, 2);
    env_int("DEJAVU_FENCED", c->fenced_env, 0, 1);
    env_int("DEJAVU_FINISH_VB", c->finish_vb_env, 0, 16);
    env_int("DEJAVU_FUSE", c->fuse_env, 0, 1);
    env_int("DEJAVU_FP4", c->fp4_env, 0, 1);
    env_int("DEJAVU_VCODE", c->vcode_env, 0, 1);
    env_int("DEJAVU_SIGNED", c->allow_signed, 0, 1);
    env_int("DEJAVU_GPAD", c->group_pad_kb, 0, 4096);
    env_int("DEJAVU_SPIN", c->spin_wait, 0, 1);
    env_int("DEJAVU_LC", c->lc_env, 0, 2);
    env_int("DEJAVU_HT", c->ht_env, 1, 2);
    env_int("DEJAVU_RING", c->ring_env, 0, 2);
    env_int("DEJAVU_MIXED", c->mixed_env, 0, 1);
    env_int("DEJAVU_TUNE_ALL", c->tune_all_env, 0, 1);
    env_int("DEJAVU_SSD_MFMA", c->ssd_mfma_env, 0, 3);
    env_int("DEJAVU_CHAINS", c->chains_env, 1, 3);
    env_int("DEJAVU_CHAIN_ORDER", c->chain_order_env, -1, 2);
    env_int("DEJAVU_LC22", c->lc22_env, 0, 1);
    env_int("DEJAVU_NT", c->nt_env, 0, 1);
    env_int("DEJAVU_TEST_FAIL_ALLOC", c->fail_alloc_env, 0, 64);
    *out = c;
    return DV_OK;
}

extern "C" void dv_destroy(dv_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    free_library(c);
    if (c->d_land) (void)hipFree(c->d_land);
    if (c->d_lut) (void)hipFree(c->d_lut);
    if (c->d_poses) (void)hipFree(c->d_poses);
    if (c->d_sense) (void)hipFree(c->d_sense);
    if (c->d_sense_err) (void)hipFree(c->d_sense_err);
    if (c->d_path) (void)hipFree(c->d_path);
    if (c->d_cover) (void)hipFree(c->d_cover);
    if (c->d_cover_slots) (void)hipFree(c->d_cover_slots);
    if (c->d_minkeys) (void)hipFree(c->d_minkeys);
    if (c->h_minkeys) (void)hipHostFree(c->h_minkeys);
    if (c->d_errstate) (void)hipFree(c->d_errstate);
    if (c->h_errout) (void)hipHostFree(c->h_errout);
    if (c->h_pub) (void)hipHostFree(c->h_pub);
    if (c->h_mbox) (void)hipHostUnregister(c->h_mbox);
    for (auto e : c->pev) (void)hipEventDestroy(e);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->aux_stream) { (void)hipStreamSynchronize(c->aux_stream); (void)hipStreamDestroy(c->aux_stream); }
    if (c->batch_stream) { (void)hipStreamSynchronize(c->batch_stream); (void)hipStreamDestroy(c->batch_stream); }
    if (c->batch_stream3) { (void)hipStreamSynchronize(c->batch_stream3); (void)hipStreamDestroy(c->batch_stream3); }
    if (c->ev_a) (void)hipEventDestroy(c->ev_a);
    if (c->ev_b) (void)hipEventDestroy(c->ev_b);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

extern "C" int dv_set_stream(dv_ctx* c, void* s) {
    if (!c) return DV_ERR_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return DV_OK;
}

extern "C" int dv_set_exact(dv_ctx* c, int exact) {
    if (!c) return DV_ERR_INVALID;
    c->exact = exact ? 1 : 0;
    return DV_OK;
}

extern "C" int dv_set_mailbox(dv_ctx* c, void* host_base, int64_t bytes, int rank, int world) {
    if (!c) return DV_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->h_mbox) { (void)hipHostUnregister(c->h_mbox); c->h_mbox = nullptr; c->d_mbox = nullptr; }
    if (!host_base) return DV_OK;                                   // detach
    const int64_t per_slot = (int64_t)world * kMboxEntry * (int64_t)sizeof(double);
    if (world < 1 || world > 64 || rank < 0 || rank >= world || bytes < per_slot || ((uintptr_t)host_base & 4095))
        return fail(c, DV_ERR_INVALID, "dv_set_mailbox: bad arguments");
    HIP_TRY(c, hipHostRegister(host_base, (size_t)bytes, hipHostRegisterMapped | hipHostRegisterPortable));
    c->h_mbox = (double*)host_base;
    HIP_TRY(c, hipHostGetDevicePointer((void**)&c->d_mbox, host_base, 0));
    c->mbox_rank = rank; c->mbox_world = world; c->mbox_slots = (int)(bytes / per_slot);
    return DV_OK;
}

extern "C" int dv_mailbox_post(dv_ctx* c, int slot, uint64_t seq) {
    if (!c) return DV_ERR_INVALID;
    if (!c->d_mbox || slot < 0 || slot >= c->mbox_slots) return fail(c, DV_ERR_STATE, "dv_mailbox_post: no mailbox or bad slot");
    if (!c->have_lib || c->A < 1) return fail(c, DV_ERR_STATE, "no library or no resident patches");
    HIP_TRY(c, hipSetDevice(c->device));
    double* entry = c->d_mbox + ((size_t)slot * c->mbox_world + c->mbox_rank) * kMboxEntry;
    hipLaunchKernelGGL(k_post, dim3(1), dim3(256), 0, c->stream, c->d_record, entry, 3 + 4 * c->A, (unsigned long long)seq);
    HIP_TRY(c, hipGetLastError());
    return DV_OK;
}

extern "C" int dv_mailbox_wait(dv_ctx* c, int slot, uint64_t seq, uint64_t rank_mask, double* records, int64_t stride,
                               int timeout_ms) {
    if (!c || !records) return DV_ERR_INVALID;
    if (!c->h_mbox || slot < 0 || slot >= c->mbox_slots || c->A < 1) return fail(c, DV_ERR_STATE, "dv_mailbox_wait: no mailbox or bad slot");
    const int n = 3 + 4 * c->A;
    if (stride < n) return fail(c, DV_ERR_INVALID, "dv_mailbox_wait: stride %lld < record size %d", (long long)stride, n);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < c->mbox_world; ++r) {
        if (!((rank_mask >> r) & 1)) continue;
        volatile const unsigned long long* e =
            reinterpret_cast<volatile const unsigned long long*>(c->h_mbox + ((size_t)slot * c->mbox_world + r) * kMboxEntry);
        double* out = records + (size_t)r * stride;
        unsigned spins = 0;
        for (;;) {
            if (e[0] == seq) {
                std::atomic_thread_fence(std::memory_order_acquire);
                unsigned long long x = 0x9E3779B97F4A7C15ull ^ (seq * 0xD1B54A32D192ED03ull);
                for (int i = 0; i < n; ++i) {
                    const unsigned long long w = e[2 + i];
                    memcpy(&out[i], &w, sizeof w);
                    x ^= w * (unsigned long long)(2 * i + 3);
                }
                if (e[0] == seq && x == e[1]) break;
            }
            if ((++spins & 1023u) == 0 && timeout_ms > 0 &&
                std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(timeout_ms))
                return fail(c, DV_ERR_STATE, "dv_mailbox_wait: rank %d has not posted sequence %llu within %d ms", r,
                            (unsigned long long)seq, timeout_ms);
        }
    }
    return DV_OK;
}

extern "C" int dv_publish(dv_ctx* c, const void* device_src, int64_t n) {
    if (!c) return DV_ERR_INVALID;
    if (!device_src || n < 1 || n > (1 << 20)) return fail(c, DV_ERR_INVALID, "dv_publish: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    if (n > c->pub_cap) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (c->h_pub) (void)hipHostFree(c->h_pub);
        c->h_pub = nullptr;
        c->pub_cap = 0;
        HIP_TRY(c, hipHostMalloc(&c->h_pub, (size_t)(n + 8) * sizeof(double), hipHostMallocMapped));
        HIP_TRY(c, hipHostGetDevicePointer((void**)&c->d_pub, c->h_pub, 0));
        memset(c->h_pub, 0, (size_t)(n + 8) * sizeof(double));
        c->pub_cap = n;
    }
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(256), 0, c->stream, (const double*)device_src, c->d_pub + 8,
                       (unsigned long long*)c->d_pub, (int)n, ++c->pub_seq);
    HIP_TRY(c, hipGetLastError());
    return DV_OK;
}

extern "C" int dv_publish_wait(dv_ctx* c, double* dst, int64_t n) {
    if (!c) return DV_ERR_INVALID;
    if (!dst || n < 1 || n > c->pub_cap || !c->h_pub) return fail(c, DV_ERR_STATE, "dv_publish_wait: nothing published");
    volatile const unsigned long long* word = (volatile const unsigned long long*)c->h_pub;
    bool seen = false;
    if (c->spin_wait) {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (!(seen = (*word == c->pub_seq))) {
            if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
        }
    }
    if (!seen) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (*word != c->pub_seq) return fail(c, DV_ERR_STATE, "dv_publish_wait: the published sequence word never arrived");
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    memcpy(dst, c->h_pub + 8, (size_t)n * sizeof(double));
    return DV_OK;
}

extern "C" int dv_synchronize(dv_ctx* c) {
    if (!c) return DV_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DV_OK;
}

// Distance between view groups in the tile array, in 16-byte units.  A group is kb = planes*Q KiB; consecutive
// groups are streamed at the same time by different waves at the same offset, so a power-of-two distance would put
// them on the same HBM channels.  DEJAVU_GPAD = KiB of padding per group (default -1: make the KiB count odd).
static long long group_stride(const dv_ctx* c, long long kb) {
    const long long pad = c->group_pad_kb >= 0 ? c->group_pad_kb : ((kb & 1) ? 0 : 1);
    return (kb + pad) * 64;
}

// ------------------------------------------------------------------ bit planes (MFMA scoring path)
// Builds the bit-plane copy of the resident byte tiles when the library's values allow it.  Never fails the ingest:
// a library that does not qualify simply keeps the byte path.
static int build_bit_planes(dv_ctx* c) {
    const LibCfg& g = c->cfg;
    c->bits_ok = false;
    c->fp4_ok = false;
    c->mixed = false;
    if (c->bits_env == 0 || c->metric != 0 || g.generic) return DV_OK;
    unsigned* d_presence = nullptr;
    uint32_t presence[(kMaxHues + 1) * 8];
    HIP_TRY(c, hipMalloc(&d_presence, sizeof presence));
    hipError_t e = hipMemsetAsync(d_presence, 0, sizeof presence, c->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_level_scan, dim3(2048), dim3(256), 0, c->stream, c->d_tiles, c->cfg, d_presence);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(presence, d_presence, sizeof presence, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_presence);
    if (e != hipSuccess) return fail(c, DV_ERR_HIP, "level scan: %s", hipGetErrorString(e));

    BitCfg b{};
    b.nbp = g.npl;
    int t = 0;
    bool mixed = false;
    for (int bp = 0; bp < g.npl; ++bp) {
        if (bp == g.nhs) b.T[0] = t;                                   // HS planes come first
        int lmin = 0, lmax = 0;
        const int n = plan_byte_plane(presence + bp * 8, kMaxBitPlanes - t, b.lo + t, b.w + t, &lmin, &lmax);
        if (n < 0) {
            // Too many levels.  A saturation plane: the MIXED layout keeps ALL saturation planes as bytes (scored with v_sad_u8 by
            // k_sad_tiles) and makes bit planes of the value plane only, scored on the matrix cores; the two passes meet in
            // k_finish (launch_int_scoring).  The value plane: byte path.
            if (bp >= g.nhs || !g.hasv || !c->mixed_env) return DV_OK;
            mixed = true;
            t = 0;
            for (int k = 0; k < g.nhs; ++k) { b.lmin[k] = 0; b.lmax[k] = 255; }
            bp = g.nhs - 1;                                             // on with the value plane
            continue;
        }
        for (int k = 0; k < n; ++k) b.pl[t + k] = (unsigned char)bp;
        b.lmin[bp] = (unsigned char)lmin;
        b.lmax[bp] = (unsigned char)lmax;
        t += n;
    }
    if (g.nhs == g.npl) b.T[0] = t;                                     // no value plane
    b.T[1] = t - b.T[0];
    const int total = t;
    if (total == 0) return DV_OK;                                       // a constant library: nothing to stream either way
    // worth it when it streams at most 3/4 of the bytes (DEJAVU_BITS=2: whenever the planes fit)
    if (c->bits_env == 1 && (total + (mixed ? 8 * g.nhs : 0)) * 4 > g.npl * 8 * 3) return DV_OK;
    // a segment's sum must fit an int32 with room to spare
    for (int seg = 0; seg < 2; ++seg)
        if ((double)b.T[seg] * g.P * 127.0 > 1.9e9) return DV_OK;
    for (int seg = 0; seg < 2; ++seg) b.NK[seg] = (int)(((long long)b.T[seg] * g.P + 255) / 256);
    // the HS K-steps in whole stages of the fp4 ring (2 or 4 K-steps): the item then runs as one loop through both segments
    // (fp4_segment's kflush); the padding K-steps hold no bits and no coefficients
    b.NK[0] = (b.NK[0] + 3) / 4 * 4;
    // and the V K-steps in whole stages of four too: the code tiles store a stage's code dwords together (k_bitpack_code)
    b.NK[1] = (b.NK[1] + 3) / 4 * 4;
    // Five value levels (four planes of the one value byte plane): the fp4 form may read them as 3-bit codes (k_bitpack_code)
    const bool five_levels = c->fp4_env != 0 && c->vcode_env != 0 && g.hasv && b.T[1] == 4 && !mixed;
    const int nkt = b.NK[0] + b.NK[1];
    b.GS = nkt | 1;
    // fp4 form of the kernel.  A gap wider than 127 was split for the int8 coefficients into planes that carry the same bits
    // (library bytes sit on levels): there the first plane stands for the whole gap and its copies for nothing.  The planes
    // that land on bit b of a nibble (K-element n = plane n % T on bit n % 4) and stand for something must share a width.
    bool one_width = true;
    for (int seg = 0; seg < 2; ++seg) {
        const int first = seg ? b.T[0] : 0, T = b.T[seg];
        const uint32_t* pres[kMaxBitPlanes];
        for (int k = 0; k < T; ++k) pres[k] = presence + (int)b.pl[first + k] * 8;
        for (int bit = 0; bit < 4; ++bit) b.wacc[seg][bit] = 0;
        if (T > 0 && !plan_fp4_segment(T, pres, b.lo + first, b.w + first, b.wfull + first, b.wacc[seg])) one_width = false;
    }
    c->fp4_ok = one_width && c->fp4_env != 0;
    if (mixed && !c->fp4_ok) return DV_OK;                              // (the mixed form has the fp4 body only)
    const long long G32 = g.Fpad / 32;
    c->btile_bytes = (size_t)G32 * nkt * 1024;
    if (hipMalloc(&c->d_btiles, (size_t)G32 * b.GS * 1024) != hipSuccess || hipMalloc(&c->d_coef, (size_t)2 * nkt * 8192) != hipSuccess) {
        (void)hipGetLastError();                                        // not enough memory for the second copy: byte path
        if (c->d_btiles) { (void)hipFree(c->d_btiles); c->d_btiles = nullptr; }
        if (c->d_coef) { (void)hipFree(c->d_coef); c->d_coef = nullptr; }
        return DV_OK;
    }
    if (c->fp4_ok && hipMalloc(&c->d_coef4, (size_t)2 * nkt * 4096) != hipSuccess) {
        (void)hipGetLastError();
        c->fp4_ok = false;                                              // the int8 form alone
    }
    // what the patch preparation needs of all this: per stored byte plane its smallest level (the patch-only terms of a
    // byte add up to |a - l_0|) and the bytes that have fp4 coefficients -- on a level, or outside the library's range
    PrepBits pb{};
    pb.enabled = 1;
    pb.fp4 = c->fp4_ok ? 1 : 0;
    for (int seg = 0; seg < 2; ++seg) { pb.T[seg] = b.T[seg]; pb.NK[seg] = b.NK[seg]; }
    for (int k = 0; k < kMaxBitPlanes; ++k)
        pb.tbl[k] = (unsigned)b.pl[k] | ((unsigned)b.lo[k] << 8) | ((unsigned)b.w[k] << 16) | ((unsigned)b.wfull[k] << 24);
    // (the images' entries past a segment's last pixel are never written by k_patch_prep: zero for good)
    HIP_TRY(c, hipMemsetAsync(c->d_coef, 0, (size_t)2 * nkt * 8192, c->stream));
    if (c->fp4_ok) HIP_TRY(c, hipMemsetAsync(c->d_coef4, 0, (size_t)2 * nkt * 4096, c->stream));
    for (int bp = 0; bp < g.npl; ++bp) {
        pb.lmin[bp] = b.lmin[bp];
        for (int v = 0; v < 256; ++v) {
            const bool level = (presence[bp * 8 + (v >> 5)] >> (v & 31)) & 1u;
            // (mixed: the saturation planes are scored as bytes, any patch byte goes)
            if ((mixed && bp < g.nhs) || level || v <= (int)b.lmin[bp] || v >= (int)b.lmax[bp]) pb.ok[bp][v >> 5] |= 1u << (v & 31);
        }
    }
    c->pbits = pb;
    c->mixed = mixed;
    b.vcode = 0;
    b.GSC = b.GS;
    if (c->fp4_ok && five_levels) {
        const long long units = 4ll * b.NK[0] + 3ll * b.NK[1];          // 256-byte units of a view group: 1-KB HS rows, 768-B V rows
        long long kib = (units + 3) / 4;
        if (kib % 2 == 0) ++kib;                                        // an odd number of KiB apart, like the bit tiles
        b.GSC = (int)(kib * 4);
        if (hipMalloc(&c->d_ctiles, (size_t)G32 * b.GSC * 256) == hipSuccess) {
            b.vcode = 1;
            c->ctile_bytes = (size_t)G32 * units * 256;
        } else {
            (void)hipGetLastError();                                    // no room for the third copy: the fp4 form reads the bit tiles
            b.GSC = b.GS;
        }
    }
    // library rows: non-temporal (used once per step).  DEJAVU_NT=0 streams them with the default policy instead -- measured equal on
    // 50 000 views x 64x64 (158 MB of bit tiles, which would fit the 256 MiB Infinity Cache between two steps): kernel 38.9-40.5 us
    // either way (tools/runs/r4_c1.sh), so that kernel is not waiting for its stream
    b.nt = c->nt_env >= 0 ? c->nt_env : 1;
    c->bcfg = b;
    const long long total_t = G32 * nkt * 64;
    hipLaunchKernelGGL(k_bitpack, dim3((unsigned)((total_t + 255) / 256)), dim3(256), 0, c->stream, c->d_tiles, c->d_btiles, c->cfg, b);
    HIP_TRY(c, hipGetLastError());
    if (b.vcode) {
        hipLaunchKernelGGL(k_bitpack_code, dim3((unsigned)((total_t + 255) / 256)), dim3(256), 0, c->stream, c->d_btiles,
                           reinterpret_cast<unsigned*>(c->d_ctiles), c->cfg, b);
        HIP_TRY(c, hipGetLastError());
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->bits_ok = true;
    c->coef_ready = false;
    return DV_OK;
}

// Coefficient image + constants of the resident patches (raw bytes in d_raw_patches), when the MFMA path may score them.
static const unsigned* offlevel_word(const dv_ctx* c) {                 // the scoring kernel's form switch for the resident patches
    return c->fp4_ok ? &c->d_acc[c->acc_parity].off : c->d_one;
}
static int enqueue_bit_prep(dv_ctx* c, bool force = false) {             // (the images are written by k_patch_prep whenever the library has bit planes)
    (void)force;
    c->coef_ready = c->bits_ok;
    return DV_OK;
}

// ------------------------------------------------------------------ library
// metric 1 / 2 (ssd_f32 / ssd_u8): the common per-step buffers only (scores, state, candidates, records); the HSV tiles, the
// byte-path operands and partial sums and the k_finish / k_fold summaries are not allocated (their ingests add their own).
static int alloc_library(dv_ctx* c, int64_t F, int h, int w, double cw, int64_t first,
                         int n_hues, const unsigned char* hues, int generic, int max_s = 255, int metric = 0) {
    free_library(c);
    c->metric = metric;
    LibCfg& g = c->cfg;
    g = LibCfg{};
    g.F = F;
    g.Fpad = (F + 63) / 64 * 64;
    g.first = first;
    g.P = h * w;
    g.Q = (g.P + 15) / 16;
    g.cw = cw;
    g.whs = 0.5 * cw;
    g.wv = 1 - cw;
    g.generic = (cw > 0.0 && generic) ? 1 : 0;
    // two hues and no saturation above 127: one signed plane instead of two one-hot planes (plane_byte)
    g.signed_s = (cw > 0.0 && !g.generic && n_hues == 2 && max_s <= 127 && c->allow_signed) ? 1 : 0;
    g.nhs = cw > 0.0 ? (g.generic ? 2 : (g.signed_s ? 1 : n_hues)) : 0;
    g.hasv = cw < 1.0 ? 1 : 0;
    g.npl = g.nhs + g.hasv;
    for (int k = 0; k < kMaxHues; ++k) g.hues[k] = (!g.generic && cw > 0.0 && k < n_hues) ? hues[k] : 0;
    c->h = h;
    c->w = w;
    // cw == 1 with an all-zero-saturation library stores nothing; keep one (zero) plane so that the
    // kernels have something to stream (it contributes |0 - 0| = 0).
    if (g.npl == 0) { g.hasv = 1; g.npl = 1; }
    g.gstride = group_stride(c, (long long)g.npl * g.Q);
    c->tile_bytes = (size_t)(g.Fpad / 64) * g.gstride * sizeof(uint4);
    const double n = (double)g.P;
    c->delta = 4.0 * (n + 8.0) * std::ldexp(1.0, -53) * n;

    if (metric == 0) {
        HIP_TRY(c, hipMalloc(&c->d_tiles, c->tile_bytes));
        HIP_TRY(c, hipMalloc(&c->d_raw_patches, (size_t)kMaxHeadings * g.P * 3));
        HIP_TRY(c, hipMalloc(&c->d_prep, (size_t)g.npl * g.Q * 4 * kMaxHeadings * sizeof(unsigned)));
    }
    HIP_TRY(c, hipMalloc(&c->d_acc, 2 * sizeof(PrepAcc)));
    HIP_TRY(c, hipMalloc(&c->d_one, sizeof(unsigned)));
    c->acc_parity = 0;
    c->pbits = PrepBits{};
    // Work items = (pixel chunk, view group).  The chunk count is chosen per launch (scoring_grid); the partial-sum
    // buffer is sized for the most chunks a launch can ask for.
    {
        const long long G = g.Fpad / 64;
        long long n = (256ll * 28 + G - 1) / G;
        if (n < 1) n = 1;
        if (n > g.Q) n = g.Q;
        if (n > 32) n = 32;
        c->nchunk_cap = (int)n;
        c->nchunk = 1;
    }
    if (metric == 0) HIP_TRY(c, hipMalloc(&c->d_part, (size_t)c->nchunk_cap * 2 * kMaxHeadings * g.Fpad * sizeof(unsigned)));
    HIP_TRY(c, hipMalloc(&c->d_pmax, (size_t)kMaxHeadings * (g.Fpad / 64) * sizeof(unsigned long long)));
    HIP_TRY(c, hipMalloc(&c->d_fam, (size_t)kMaxHeadings * g.Fpad * sizeof(double)));
    HIP_TRY(c, hipMalloc(&c->d_scene, (size_t)g.Fpad * sizeof(double)));
    // per-agent state of a batched pass: up to kMaxHeadings agents (one heading each)
    HIP_TRY(c, hipMalloc(&c->d_state, kMaxHeadings * sizeof(StepState)));
    // summaries per agent: one per 256 views (k_finish) or one per workgroup of a fused scoring pass (at most 256: launch_mfma_dual_f)
    if (metric == 0) {
        HIP_TRY(c, hipMalloc(&c->d_bsum, (size_t)std::max<long long>((g.F + 255) / 256, 256) * 2 * kMaxHeadings * sizeof(unsigned long long)));
        HIP_TRY(c, hipMalloc(&c->d_bsum2, (size_t)kFoldSlices * 2 * kMaxHeadings * sizeof(unsigned long long)));
        HIP_TRY(c, hipMalloc(&c->d_ctmp, (size_t)kMaxHeadings * kTmpCap * 2 * sizeof(unsigned long long)));
    }
    HIP_TRY(c, hipMalloc(&c->d_cand, (size_t)kMaxHeadings * kCandCap * sizeof(unsigned long long)));
    HIP_TRY(c, hipMalloc(&c->d_cand_exact, (size_t)kMaxHeadings * kCandCap * sizeof(double)));
    HIP_TRY(c, hipMalloc(&c->d_record, (size_t)kMaxHeadings * (3 + 4 * kMaxHeadings) * sizeof(double)));
    HIP_TRY(c, hipMalloc(&c->d_keys, (size_t)(kMaxHeadings + kKeyWordsPerRank * 64) * sizeof(unsigned long long)));
    HIP_TRY(c, hipHostMalloc(&c->h_result, kMaxHeadings * sizeof(StepResultDev), hipHostMallocMapped));
    HIP_TRY(c, hipHostGetDevicePointer((void**)&c->d_result, c->h_result, 0));
    memset(c->h_result, 0, kMaxHeadings * sizeof(StepResultDev));
    HIP_TRY(c, hipHostMalloc(&c->h_scene, (size_t)g.Fpad * sizeof(double)));
    HIP_TRY(c, hipMemsetAsync(c->d_acc, 0, 2 * sizeof(PrepAcc), c->stream));
    {
        static const unsigned one = 1u;
        HIP_TRY(c, hipMemcpyAsync(c->d_one, &one, sizeof one, hipMemcpyHostToDevice, c->stream));
    }
    HIP_TRY(c, hipMemsetAsync(c->d_state, 0, kMaxHeadings * sizeof(StepState), c->stream));
    c->have_lib = true;
    return DV_OK;
}

static int check_lib_args(dv_ctx* c, int64_t F, int h, int w, double cw) {
    if (!c) return DV_ERR_INVALID;
    if (F < 1 || h < 1 || w < 1) return fail(c, DV_ERR_INVALID, "library needs n_views, h, w >= 1 (got %lld, %d, %d)", (long long)F, h, w);
    if (!(cw >= 0.0 && cw <= 1.0)) return fail(c, DV_ERR_INVALID, "chem_weight %g outside [0, 1]", cw);
    if ((int64_t)h * w > (1 << 22)) return fail(c, DV_ERR_INVALID, "sensor of %d x %d pixels is too large", h, w);
    if (F >= (1ll << 40)) return fail(c, DV_ERR_INVALID, "too many views");
    return DV_OK;
}

// Library ingest from a raw uint8[F][h*w][3] buffer already on the device: hue scan, layout choice, re-tile.
static int ingest_raw(dv_ctx* c, const unsigned char* d_raw, int64_t F, int h, int w, double cw, int64_t first) {
    unsigned bitmap[9] = {0};
    if (cw > 0.0) {
        unsigned* d_bitmap = nullptr;
        hipError_t e = hipMalloc(&d_bitmap, sizeof bitmap);
        if (e == hipSuccess) e = hipMemsetAsync(d_bitmap, 0, sizeof bitmap, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_hue_scan, dim3(1024), dim3(256), 0, c->stream, d_raw, (long long)F * h * w, d_bitmap);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(bitmap, d_bitmap, sizeof bitmap, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (d_bitmap) (void)hipFree(d_bitmap);
        if (e != hipSuccess) return fail(c, DV_ERR_HIP, "hue scan: %s", hipGetErrorString(e));
    }
    unsigned char hues[256];
    int n_hues = 0;
    for (int v = 0; v < 256; ++v)
        if (bitmap[v >> 5] & (1u << (v & 31))) hues[n_hues++] = (unsigned char)v;
    const int generic = n_hues > kMaxHues;

    int rc = alloc_library(c, F, h, w, cw, first, n_hues, hues, generic, (int)bitmap[8]);
    if (rc) { free_library(c); return rc; }
    const long long total = (c->cfg.Fpad / 64) * (long long)c->cfg.npl * c->cfg.Q * 64;
    hipLaunchKernelGGL(k_retile, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_raw, c->d_tiles, c->cfg);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) { free_library(c); return fail(c, DV_ERR_HIP, "retile: %s", hipGetErrorString(e)); }
    rc = build_bit_planes(c);
    if (rc) { free_library(c); return rc; }
    return DV_OK;
}

static int enqueue_sense(dv_ctx* c, const double* x, const double* y, const double* angle, long long n, unsigned char* d_out);
static int check_sense_error(dv_ctx* c);
static int ensure_sense_buffer(dv_ctx* c, size_t bytes);

// Appends n raw views (device buffer uint8[n][h*w][3]) to the resident library: the byte tiles of the existing views
// are copied as they are, only the new view groups are re-tiled; the per-step buffers are re-sized; the workgroup
// shapes timed on the old library are kept.  The device layout was chosen from the first ingest's hue set and
// saturation range: views that do not fit it cannot be appended (DV_ERR_STATE; re-ingest the whole library).
static int append_raw(dv_ctx* c, const unsigned char* d_raw, int64_t n) {
    const LibCfg old = c->cfg;
    if (old.cw > 0.0 && !old.generic) {
        unsigned bitmap[9] = {0};
        unsigned* d_bitmap = nullptr;
        hipError_t e = hipMalloc(&d_bitmap, sizeof bitmap);
        if (e == hipSuccess) e = hipMemsetAsync(d_bitmap, 0, sizeof bitmap, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_hue_scan, dim3(256), dim3(256), 0, c->stream, d_raw, (long long)n * old.P, d_bitmap);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(bitmap, d_bitmap, sizeof bitmap, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (d_bitmap) (void)hipFree(d_bitmap);
        if (e != hipSuccess) return fail(c, DV_ERR_HIP, "hue scan: %s", hipGetErrorString(e));
        const int nk = old.signed_s ? 2 : old.nhs;
        for (int v = 0; v < 256; ++v) {
            if (!(bitmap[v >> 5] & (1u << (v & 31)))) continue;
            bool known = false;
            for (int k = 0; k < nk; ++k) known |= (old.hues[k] == v);
            if (!known) return fail(c, DV_ERR_STATE, "appended views contain hue %d, which the resident layout has no plane for: "
                                    "re-ingest the whole library", v);
        }
        if (old.signed_s && bitmap[8] > 127)
            return fail(c, DV_ERR_STATE, "appended views contain saturation %u > 127, which the resident signed-saturation plane "
                        "cannot hold: re-ingest the whole library", bitmap[8]);
    }
    uint4* old_tiles = c->d_tiles;
    const size_t old_bytes = (size_t)(old.Fpad / 64) * old.gstride * sizeof(uint4);
    c->d_tiles = nullptr;                                  // survives the re-allocation below
    int tuned[4];
    for (int i = 0; i < 4; ++i) tuned[i] = c->tuned_shape[i];
    const int nk = old.signed_s ? 2 : old.nhs;
    int rc = alloc_library(c, old.F + n, c->h, c->w, old.cw, old.first, (old.cw > 0.0 && !old.generic) ? nk : 0, old.hues,
                           old.generic, old.signed_s ? 127 : 255);
    if (rc) { (void)hipFree(old_tiles); free_library(c); return rc; }
    for (int i = 0; i < 4; ++i) c->tuned_shape[i] = tuned[i];
    const LibCfg& g = c->cfg;
    if (g.npl != old.npl || g.gstride != old.gstride || g.signed_s != old.signed_s || g.nhs != old.nhs) {
        (void)hipFree(old_tiles);
        free_library(c);
        return fail(c, DV_ERR_STATE, "internal: the grown library chose another layout");
    }
    hipError_t e = hipMemcpyAsync(c->d_tiles, old_tiles, old_bytes, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) {
        const long long total = (g.Fpad / 64 - old.F / 64) * (long long)g.npl * g.Q * 64;
        hipLaunchKernelGGL(k_retile_append, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_raw, c->d_tiles, c->cfg,
                           (long long)old.F);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(old_tiles);
    if (e != hipSuccess) { free_library(c); return fail(c, DV_ERR_HIP, "append: %s", hipGetErrorString(e)); }
    rc = build_bit_planes(c);
    if (rc) { free_library(c); return rc; }
    return DV_OK;
}

extern "C" int dv_append_library(dv_ctx* c, const uint8_t* views, int64_t n, int channels) {
    if (!c) return DV_ERR_INVALID;
    if (!c->have_lib || c->metric != 0) return fail(c, DV_ERR_STATE, "no sads_hsv library to append to (call dv_set_library first)");
    if (!views || n < 1) return fail(c, DV_ERR_INVALID, "dv_append_library: views is NULL or n < 1");
    if (channels != 3) return fail(c, DV_ERR_INVALID, "views must have 3 channels (H,S,V), got %d", channels);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t raw_bytes = (size_t)n * c->cfg.P * 3;
    unsigned char* d_raw = nullptr;
    HIP_TRY(c, hipMalloc(&d_raw, raw_bytes));
    int rc = DV_OK;
    if (hipMemcpyAsync(d_raw, views, raw_bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess)
        rc = fail(c, DV_ERR_HIP, "upload: %s", hipGetErrorString(hipGetLastError()));
    if (rc == DV_OK) rc = append_raw(c, d_raw, n);
    (void)hipFree(d_raw);
    return rc;
}

extern "C" int dv_append_library_from_poses(dv_ctx* c, const double* x, const double* y, const double* angle, int64_t n,
                                            uint8_t* out_views) {
    if (!c || !x || !y || !angle || n < 1) return DV_ERR_INVALID;
    if (!c->have_lib || c->metric != 0) return fail(c, DV_ERR_STATE, "no sads_hsv library to append to");
    if (!c->have_sensor) return fail(c, DV_ERR_STATE, "sensor not configured");
    if (c->sensor.sw != c->w || c->sensor.sh != c->h) return fail(c, DV_ERR_STATE, "sensor and library shapes differ");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const size_t bytes = (size_t)n * c->sensor.sh * c->sensor.sw * 3;
    int rc = ensure_sense_buffer(c, bytes);
    if (rc) return rc;
    rc = enqueue_sense(c, x, y, angle, n, c->d_sense);
    if (rc) return rc;
    if (out_views) HIP_TRY(c, hipMemcpyAsync(out_views, c->d_sense, bytes, hipMemcpyDeviceToHost, c->stream));
    rc = check_sense_error(c);
    if (rc) return rc;
    return append_raw(c, c->d_sense, n);
}

extern "C" int dv_set_library(dv_ctx* c, const uint8_t* views, int64_t F, int h, int w, int channels,
                              double cw, int64_t first) {
    int rc = check_lib_args(c, F, h, w, cw);
    if (rc) return rc;
    if (!views) return fail(c, DV_ERR_INVALID, "views is NULL");
    if (channels != 3) return fail(c, DV_ERR_INVALID, "views must have 3 channels (H,S,V), got %d", channels);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_library(c);
    const size_t raw_bytes = (size_t)F * h * w * 3;
    unsigned char* d_raw = nullptr;
    HIP_TRY(c, hipMalloc(&d_raw, raw_bytes));
    hipError_t e = hipMemcpyAsync(d_raw, views, raw_bytes, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { (void)hipFree(d_raw); return fail(c, DV_ERR_HIP, "upload: %s", hipGetErrorString(e)); }
    rc = ingest_raw(c, d_raw, F, h, w, cw, first);
    (void)hipFree(d_raw);
    return rc;
}

// ------------------------------------------------------------------ ssd_f32 metric
// Allocations of the ssd_f32 / ssd_u8 ingests behind alloc_library: any failure frees the whole half-built library (the
// context is then without a library: the step calls answer DV_ERR_STATE) and reports DV_ERR_OOM / DV_ERR_HIP.
template <class T>
static hipError_t lib_malloc(dv_ctx* c, T** p, size_t bytes) {
    if (c->fail_alloc_env > 0 && ++c->lib_allocs == c->fail_alloc_env) { *p = nullptr; return hipErrorOutOfMemory; }
    return hipMalloc(p, bytes);
}
static int lib_fail(dv_ctx* c, hipError_t e, const char* what) {
    (void)hipGetLastError();
    free_library(c);
    return fail(c, e == hipErrorOutOfMemory ? DV_ERR_OOM : DV_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}
static int alloc_f32_buffers(dv_ctx* c) {
    const LibCfg& g = c->cfg;
    c->lib_allocs = 0;
    hipError_t e = lib_malloc(c, &c->d_ftiles, c->tile_bytes);
    if (e == hipSuccess) e = lib_malloc(c, &c->d_fraw, (size_t)kMaxHeadings * g.P * sizeof(float));
    if (e == hipSuccess) e = lib_malloc(c, &c->d_fprep, (size_t)g.Q * 4 * kMaxHeadings * sizeof(float));
    if (e == hipSuccess) e = lib_malloc(c, &c->d_fpart, (size_t)c->nchunk_cap * kMaxHeadings * g.Fpad * sizeof(double));
    if (e == hipSuccess) e = lib_malloc(c, &c->d_fprep4, (size_t)g.Q * kMaxHeadings * sizeof(float4));
    if (e == hipSuccess) e = lib_malloc(c, &c->d_fvnorm, (size_t)g.Fpad * sizeof(double));
    if (e == hipSuccess) e = lib_malloc(c, &c->d_fpnorm, (size_t)kMaxHeadings * sizeof(double));
    if (e == hipSuccess) e = lib_malloc(c, &c->d_flower, (size_t)kMaxHeadings * kF32xShards * sizeof(unsigned long long));
    if (e == hipSuccess) e = lib_malloc(c, &c->d_fprepb, (size_t)2 * ((g.P + 15) / 16) * 2 * 64 * sizeof(uint4));
    if (e != hipSuccess) return lib_fail(c, e, "ssd_f32 buffers");
    return DV_OK;
}

static void negate_result(dv_step_result* r) {
    r->best_fam = -r->best_fam;
    r->approx_max = -r->approx_max;
    for (int a = 0; a < r->n_headings; ++a) {
        r->angle_fam[a] = -r->angle_fam[a];
        if (r->exact_view[a] >= 0) r->exact_fam[a] = -r->exact_fam[a];
    }
}

// Common buffers, layout constants and the ssd_f32 buffers of a library of F float32 views (no views yet).
static int prepare_f32_library(dv_ctx* c, int64_t F, int h, int w, int64_t first) {
    int rc = alloc_library(c, F, h, w, 0.0, first, 0, nullptr, 0, 255, 1);       // common per-step buffers (fam, state, records)
    if (rc) { free_library(c); return rc; }
    LibCfg& g = c->cfg;
    g.Q = (g.P + 3) / 4;                                              // 4 float pixels per 16-byte chunk
    g.npl = 1; g.nhs = 0; g.hasv = 1;
    g.gstride = group_stride(c, (long long)g.Q);
    c->tile_bytes = (size_t)(g.Fpad / 64) * g.gstride * sizeof(float4);
    {
        const long long G = g.Fpad / 64;
        long long n = (256ll * 28 + G - 1) / G;
        const long long q4 = (g.Q + 3) / 4;
        if (n > q4) n = q4;
        if (n > 16) n = 16;
        if (n < 1) n = 1;
        c->nchunk_cap = (int)n;
    }
    // fp32 accumulation over 8 pixels, then double: relative error of a score <= 1e-6 (bound), typically 1e-7
    c->delta = 0.0;
    return alloc_f32_buffers(c);
}

extern "C" int dv_set_library_f32(dv_ctx* c, const float* views, int64_t F, int h, int w, int64_t first) {
    int rc = check_lib_args(c, F, h, w, 0.0);
    if (rc) return rc;
    if (!views) return fail(c, DV_ERR_INVALID, "views is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    rc = prepare_f32_library(c, F, h, w, first);
    if (rc) return rc;
    const LibCfg& g = c->cfg;
    float* d_raw = nullptr;
    hipError_t e = lib_malloc(c, &d_raw, (size_t)F * g.P * sizeof(float));
    if (e == hipSuccess) e = hipMemcpyAsync(d_raw, views, (size_t)F * g.P * sizeof(float), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        const long long total = (g.Fpad / 64) * (long long)g.Q * 64;
        hipLaunchKernelGGL(k_retile_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_raw, c->d_ftiles, c->cfg);
        hipLaunchKernelGGL(k_norm_f32, dim3((unsigned)(g.Fpad / 64)), dim3(64), 0, c->stream, c->d_ftiles, c->d_fvnorm, c->cfg);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (d_raw) (void)hipFree(d_raw);
    if (e != hipSuccess) return lib_fail(c, e, "ssd_f32 ingest");
    return DV_OK;
}

extern "C" int dv_generate_library_f32(dv_ctx* c, uint64_t seed, int64_t F, int h, int w, int64_t first) {
    int rc = check_lib_args(c, F, h, w, 0.0);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    rc = prepare_f32_library(c, F, h, w, first);
    if (rc) return rc;
    const long long total = (c->cfg.Fpad / 64) * (long long)c->cfg.Q * 64;
    hipLaunchKernelGGL(k_generate_tiles_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->d_ftiles, c->cfg,
                       (unsigned long long)seed);
    hipLaunchKernelGGL(k_norm_f32, dim3((unsigned)(c->cfg.Fpad / 64)), dim3(64), 0, c->stream, c->d_ftiles, c->d_fvnorm, c->cfg);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return lib_fail(c, e, "ssd_f32 generator");
    return DV_OK;
}

static int upload_patches_f32(dv_ctx* c, const float* patches, int A) {
    if (!c) return DV_ERR_INVALID;
    if (!c->have_lib || c->metric != 1) return fail(c, DV_ERR_STATE, "no ssd_f32 library set (call dv_set_library_f32 first)");
    if (A < 1 || A > kMaxHeadings) return fail(c, DV_ERR_INVALID, "ssd_f32: n_headings %d outside [1, %d]", A, kMaxHeadings);
    if (!patches) return fail(c, DV_ERR_INVALID, "patches is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_fraw, patches, (size_t)A * c->cfg.P * sizeof(float), hipMemcpyHostToDevice, c->stream));
    c->A = A; c->n_agents = 1; c->A_agent = A; c->patches_sensed = false;
    c->APAD = A <= 8 ? 8 : (A <= 16 ? 16 : (A <= 32 ? 32 : 64));
    // the matrix-core form's operand rows, patch norms and cleared lower bounds in one launch; the direct form's operand layout
    // only when a direct pass is about to run (ensure_direct_prep_f32: dv_score_f32, scene_ssd, DEJAVU_SSD_MFMA=0)
    c->fprep_direct_ready = false;
    if (c->ssd_mfma_env) {
        const long long t4 = (long long)c->cfg.Q * c->APAD;
        const int nrow = (int)((t4 + 255) / 256);
        hipLaunchKernelGGL(k_prep_f32x, dim3((unsigned)(nrow + A)), dim3(256), 0, c->stream, c->d_fraw, c->d_fprep4, c->d_fpnorm, c->d_flower,
                           c->cfg, A, c->APAD, nrow);
        if ((c->APAD > 16 && c->ssd_mfma_env == 1) || c->ssd_mfma_env == 3) {      // the two-term bf16 form's operand rows (passes of 32 headings)
            const int passes = c->APAD > 32 ? 2 : 1;
            const long long tb = (long long)passes * ((c->cfg.P + 15) / 16) * 64;
            hipLaunchKernelGGL(k_prep_f32b, dim3((unsigned)((tb + 255) / 256)), dim3(256), 0, c->stream, c->d_fraw, c->d_fprepb, c->cfg, A, passes);
        }
        HIP_TRY(c, hipGetLastError());
    }
    return DV_OK;
}

static int ensure_direct_prep_f32(dv_ctx* c) {
    if (c->fprep_direct_ready) return DV_OK;
    const long long total = (long long)c->cfg.Q * 4 * c->APAD;
    hipLaunchKernelGGL(k_prep_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->d_fraw, c->d_fprep, c->cfg, c->A, c->APAD);
    HIP_TRY(c, hipGetLastError());
    c->fprep_direct_ready = true;
    return DV_OK;
}

static int enqueue_step(dv_ctx* c, uint32_t flags, bool want_scene);
static void launch_fold(dv_ctx* c, int nb, StepResultDev* outp, double* recp, int force, int seq, const unsigned long long* serr);
static const unsigned long long* sense_err_ptr(const dv_ctx* c);
static bool batch_pass_fuses(dv_ctx* c, int apad);
static int wait_step(dv_ctx* c, dv_step_result* result, double* scene_fam);
static int finish_pass(dv_ctx* c);
static void copy_result(const dv_ctx* c, int agent, dv_step_result* result);
static bool spin_for_records(dv_ctx* c, int slot, int n, int A, int seq);

extern "C" int dv_step_f32(dv_ctx* c, const float* patches, int A, uint32_t flags, dv_step_result* result, double* scene_ssd) {
    int rc = upload_patches_f32(c, patches, A);
    if (rc) return rc;
    if (!result) return fail(c, DV_ERR_INVALID, "result is NULL");
    rc = enqueue_step(c, flags, scene_ssd != nullptr);
    if (rc) return rc;
    rc = wait_step(c, result, scene_ssd);
    if (rc) return rc;
    negate_result(result);
    if (scene_ssd) for (int64_t f = 0; f < c->cfg.F; ++f) scene_ssd[f] = -scene_ssd[f];
    return DV_OK;
}

// ------------------------------------------------------------------ ssd_u8 metric
// Exact SSD of single-channel uint8 views on the int8 matrix cores (k_ssd_u8_mfma); results as ssd_f32's.
// Ingest of a raw single-channel library already on the device (uint8[F][h*w]); frees nothing of the caller's.
static int ingest_u8(dv_ctx* c, const unsigned char* d_raw, int64_t F, int h, int w, int64_t first) {
    int rc = alloc_library(c, F, h, w, 0.0, first, 0, nullptr, 0, 255, 2);       // common per-step buffers (fam, state, records)
    if (rc) { free_library(c); return rc; }
    LibCfg& g = c->cfg;
    g.npl = 1; g.nhs = 0; g.hasv = 1;
    const int K = (g.P + 31) / 32;
    c->u8_K = K;
    c->u8_KC = K < 128 ? K : 128;                                       // 1 KB of LDS per K-step: at most 128 KB of operand rows per chunk
    c->u8_nchunk = (K + c->u8_KC - 1) / c->u8_KC;
    c->u8_KC = (K + c->u8_nchunk - 1) / c->u8_nchunk;                  // (chunks of equal length)
    c->nchunk_cap = c->u8_nchunk;
    c->tile_bytes = (size_t)(g.Fpad / 32) * K * 1024;
    c->lib_allocs = 0;
    hipError_t e = lib_malloc(c, &c->d_u8tiles, c->tile_bytes);
    if (e == hipSuccess) e = lib_malloc(c, &c->d_u8raw, (size_t)kMaxHeadings * g.P);
    if (e == hipSuccess) e = lib_malloc(c, &c->d_u8prep, (size_t)2 * K * 1024);
    if (e == hipSuccess) e = lib_malloc(c, &c->d_u8part, (size_t)c->u8_nchunk * kMaxHeadings * g.Fpad * sizeof(int));
    if (e == hipSuccess) e = lib_malloc(c, &c->d_vnorm, (size_t)g.Fpad * sizeof(unsigned long long));
    if (e == hipSuccess) e = lib_malloc(c, &c->d_pnorm, (size_t)kMaxHeadings * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemsetAsync(c->d_vnorm, 0, (size_t)g.Fpad * sizeof(unsigned long long), c->stream);
    if (e == hipSuccess) {
        const long long total = (g.Fpad / 32) * (long long)K * 64;
        hipLaunchKernelGGL(k_retile_u8, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, d_raw, c->d_u8tiles, c->d_vnorm, c->cfg, K);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return lib_fail(c, e, "ssd_u8 ingest");
    c->delta = 0.0;                                                     // exact integers: ties go by index (k_tail's exact rule)
    return DV_OK;
}

extern "C" int dv_set_library_u8(dv_ctx* c, const uint8_t* views, int64_t F, int h, int w, int64_t first) {
    int rc = check_lib_args(c, F, h, w, 0.0);
    if (rc) return rc;
    if (!views) return fail(c, DV_ERR_INVALID, "views is NULL");
    if ((long long)h * w > 131071) return fail(c, DV_ERR_INVALID, "ssd_u8: %d x %d pixels: the int32 cross terms hold at most 131071", h, w);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_library(c);
    unsigned char* d_raw = nullptr;
    HIP_TRY(c, hipMalloc(&d_raw, (size_t)F * h * w));
    hipError_t e = hipMemcpyAsync(d_raw, views, (size_t)F * h * w, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) { (void)hipFree(d_raw); return fail(c, DV_ERR_HIP, "upload: %s", hipGetErrorString(e)); }
    rc = ingest_u8(c, d_raw, F, h, w, first);
    (void)hipFree(d_raw);
    return rc;
}

// Operand rows and norms of the A resident single-channel patches (d_u8raw) for the matrix-core pass.
static int prep_patches_u8(dv_ctx* c, int A) {
    HIP_TRY(c, hipMemsetAsync(c->d_pnorm, 0, (size_t)kMaxHeadings * sizeof(unsigned long long), c->stream));
    c->A = A; c->n_agents = 1; c->A_agent = A;
    c->APAD = A <= 32 ? 32 : 64;
    const int passes = c->APAD / 32;
    const long long total = (long long)passes * c->u8_K * 64;
    hipLaunchKernelGGL(k_prep_u8, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->d_u8raw, c->d_u8prep, c->d_pnorm, c->cfg, c->u8_K, A,
                       passes);
    HIP_TRY(c, hipGetLastError());
    return DV_OK;
}

static int upload_patches_u8(dv_ctx* c, const uint8_t* patches, int A) {
    if (!c) return DV_ERR_INVALID;
    if (!c->have_lib || c->metric != 2) return fail(c, DV_ERR_STATE, "no ssd_u8 library set (call dv_set_library_u8 first)");
    if (A < 1 || A > kMaxHeadings) return fail(c, DV_ERR_INVALID, "ssd_u8: n_headings %d outside [1, %d]", A, kMaxHeadings);
    if (!patches) return fail(c, DV_ERR_INVALID, "patches is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_u8raw, patches, (size_t)A * c->cfg.P, hipMemcpyHostToDevice, c->stream));
    c->patches_sensed = false;
    return prep_patches_u8(c, A);
}

extern "C" int dv_step_u8(dv_ctx* c, const uint8_t* patches, int A, uint32_t flags, dv_step_result* result, double* scene_ssd) {
    int rc = upload_patches_u8(c, patches, A);
    if (rc) return rc;
    if (!result) return fail(c, DV_ERR_INVALID, "result is NULL");
    rc = enqueue_step(c, flags, scene_ssd != nullptr);
    if (rc) return rc;
    rc = wait_step(c, result, scene_ssd);
    if (rc) return rc;
    negate_result(result);
    if (scene_ssd) for (int64_t f = 0; f < c->cfg.F; ++f) scene_ssd[f] = -scene_ssd[f];
    return DV_OK;
}

// The ssd_u8 metric behind the sensor model (the agent's step when SSD is its familiarity plug-in): train_from_path and the
// heading loop on the device, the compared channel taken out of the sensed HSV bytes by k_take_channel.
extern "C" int dv_set_library_u8_from_poses(dv_ctx* c, const double* x, const double* y, const double* angle, int64_t n, int channel,
                                            int64_t first, uint8_t* out_views) {
    if (!c || !x || !y || !angle) return DV_ERR_INVALID;
    if (!c->have_sensor) return fail(c, DV_ERR_STATE, "sensor not configured");
    if (channel < 0 || channel > 2) return fail(c, DV_ERR_INVALID, "channel %d outside [0, 2]", channel);
    int rc = check_lib_args(c, n, c->sensor.sh, c->sensor.sw, 0.0);
    if (rc) return rc;
    if ((long long)c->sensor.sh * c->sensor.sw > 131071) return fail(c, DV_ERR_INVALID, "ssd_u8: at most 131071 pixels per view");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_library(c);
    const long long n_px = (long long)n * c->sensor.sh * c->sensor.sw;
    rc = ensure_sense_buffer(c, (size_t)n_px * 3);
    if (rc) return rc;
    rc = enqueue_sense(c, x, y, angle, n, c->d_sense);
    if (rc) return rc;
    if (out_views) HIP_TRY(c, hipMemcpyAsync(out_views, c->d_sense, (size_t)n_px * 3, hipMemcpyDeviceToHost, c->stream));
    rc = check_sense_error(c);
    if (rc) return rc;
    unsigned char* d_raw = nullptr;
    HIP_TRY(c, hipMalloc(&d_raw, (size_t)n_px));
    hipLaunchKernelGGL(k_take_channel, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, c->stream, c->d_sense, d_raw, n_px, channel);
    if (hipGetLastError() != hipSuccess) { (void)hipFree(d_raw); return fail(c, DV_ERR_HIP, "k_take_channel launch failed"); }
    rc = ingest_u8(c, d_raw, n, c->sensor.sh, c->sensor.sw, first);
    (void)hipFree(d_raw);
    return rc;
}

extern "C" int dv_sense_step_u8(dv_ctx* c, double x, double y, const double* angles, int A, int channel, uint32_t flags,
                                dv_step_result* result, double* scene_ssd) {
    if (!c) return DV_ERR_INVALID;
    if (!c->have_lib || c->metric != 2) return fail(c, DV_ERR_STATE, "no ssd_u8 library set (call dv_set_library_u8 first)");
    if (A < 1 || A > kMaxHeadings) return fail(c, DV_ERR_INVALID, "ssd_u8: n_headings %d outside [1, %d]", A, kMaxHeadings);
    if (channel < 0 || channel > 2) return fail(c, DV_ERR_INVALID, "channel %d outside [0, 2]", channel);
    if (!angles || !result) return fail(c, DV_ERR_INVALID, "angles or result is NULL");
    if (!c->have_sensor) return fail(c, DV_ERR_STATE, "sensor not configured");
    if (c->sensor.sw != c->w || c->sensor.sh != c->h)
        return fail(c, DV_ERR_STATE, "sensor is %dx%d but the library holds %dx%d views", c->sensor.sw, c->sensor.sh, c->w, c->h);
    HIP_TRY(c, hipSetDevice(c->device));
    const long long n_px = (long long)A * c->cfg.P;
    int rc = ensure_sense_buffer(c, (size_t)kMaxHeadings * c->cfg.P * 3);      // (sized once for any heading count: no allocation in later steps)
    if (rc) return rc;
    double xs[kMaxHeadings], ys[kMaxHeadings];
    for (int a = 0; a < A; ++a) { xs[a] = x; ys[a] = y; }
    rc = enqueue_sense(c, xs, ys, angles, A, c->d_sense);
    if (rc) return rc;
    hipLaunchKernelGGL(k_take_channel, dim3((unsigned)((n_px + 255) / 256)), dim3(256), 0, c->stream, c->d_sense, c->d_u8raw, n_px, channel);
    HIP_TRY(c, hipGetLastError());
    c->patches_sensed = true;                                  // k_tail reads k_sense's error word (sense_err_ptr)
    rc = prep_patches_u8(c, A);
    if (rc) return rc;
    rc = enqueue_step(c, flags, scene_ssd != nullptr);
    if (rc) return rc;
    rc = wait_step(c, result, scene_ssd);
    if (rc) return rc;
    negate_result(result);
    if (scene_ssd) for (int64_t f = 0; f < c->cfg.F; ++f) scene_ssd[f] = -scene_ssd[f];
    return DV_OK;
}

// ------------------------------------------------------------------ sensor model
static int check_step_args(dv_ctx* c, int A);
static int prep_patches(dv_ctx* c, int A);

extern "C" int dv_set_landscape(dv_ctx* c, const uint8_t* landscape, int rows, int cols, int channels) {
    if (!c || !landscape) return DV_ERR_INVALID;
    if (rows < 1 || cols < 1 || channels != 3) return fail(c, DV_ERR_INVALID, "landscape must be uint8[rows, cols, 3]");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_land) { (void)hipFree(c->d_land); c->d_land = nullptr; }
    HIP_TRY(c, hipMalloc(&c->d_land, (size_t)rows * cols * 3));
    HIP_TRY(c, hipMemcpyAsync(c->d_land, landscape, (size_t)rows * cols * 3, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->sensor.rows = rows;
    c->sensor.cols = cols;
    return DV_OK;
}

extern "C" int dv_configure_sensor(dv_ctx* c, int sw, int sh, int pw, int ph, const uint8_t* lut, int mask_n) {
    if (!c || !lut) return DV_ERR_INVALID;
    if (!c->d_land) return fail(c, DV_ERR_STATE, "no landscape set (call dv_set_landscape first)");
    if (sw < 1 || sh < 1 || pw < 1 || ph < 1 || pw * ph > 4096) return fail(c, DV_ERR_INVALID, "bad sensor geometry");
    if (mask_n < 0 || mask_n > sw / 2) return fail(c, DV_ERR_INVALID, "mask_middle_n %d outside [0, %d]", mask_n, sw / 2);
    HIP_TRY(c, hipSetDevice(c->device));
    if (!c->d_lut) HIP_TRY(c, hipMalloc(&c->d_lut, 768));
    if (!c->d_sense_err) {
        HIP_TRY(c, hipMalloc(&c->d_sense_err, sizeof(unsigned long long)));
        HIP_TRY(c, hipMemsetAsync(c->d_sense_err, 0, sizeof(unsigned long long), c->stream));
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_lut, lut, 768, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->sensor.sw = sw; c->sensor.sh = sh; c->sensor.pw = pw; c->sensor.ph = ph; c->sensor.mask_n = mask_n;
    c->have_sensor = true;
    return DV_OK;
}

// Enqueue k_sense for n poses into `d_out` (uint8[n][sh][sw][3]).  Rotation cos/sin come from the host's libm,
// like the reference's cimported cos/sin (util.pyx:143-145).
static int enqueue_sense(dv_ctx* c, const double* x, const double* y, const double* angle, long long n, unsigned char* d_out) {
    if ((size_t)n > c->poses_cap) {
        if (c->d_poses) (void)hipFree(c->d_poses);
        c->d_poses = nullptr;
        c->poses_cap = 0;
        HIP_TRY(c, hipMalloc(&c->d_poses, (size_t)n * sizeof(Pose)));
        c->poses_cap = (size_t)n;
    }
    c->h_poses.resize((size_t)n);
    for (long long i = 0; i < n; ++i) {
        const double rot = -(0.5 * M_PI - angle[i]);
        c->h_poses[(size_t)i] = Pose{x[i], y[i], std::cos(rot), std::sin(rot)};
    }
    HIP_TRY(c, hipMemsetAsync(c->d_sense_err, 0, sizeof(unsigned long long), c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_poses, c->h_poses.data(), (size_t)n * sizeof(Pose), hipMemcpyHostToDevice, c->stream));
    const long long total = n * c->sensor.sh * c->sensor.sw;
    hipLaunchKernelGGL(k_sense, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->d_land, c->d_poses, (int)n,
                       c->sensor, c->d_lut, d_out, reinterpret_cast<int*>(c->d_sense_err));
    HIP_TRY(c, hipGetLastError());
    return DV_OK;
}

static int check_sense_error(dv_ctx* c) {
    int err = 0;
    HIP_TRY(c, hipMemcpyAsync(&err, c->d_sense_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));   // k_sense's flag: low word
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (err) return fail(c, DV_ERR_INDEX, "sensor footprint reaches past the end of the landscape (index out of bounds)");
    return DV_OK;
}

static int ensure_sense_buffer(dv_ctx* c, size_t bytes) {
    if (bytes > c->sense_cap) {
        if (c->d_sense) (void)hipFree(c->d_sense);
        c->d_sense = nullptr;
        c->sense_cap = 0;
        HIP_TRY(c, hipMalloc(&c->d_sense, bytes));
        c->sense_cap = bytes;
    }
    return DV_OK;
}

extern "C" int dv_sense(dv_ctx* c, const double* x, const double* y, const double* angle, int n, uint8_t* out) {
    if (!c || !x || !y || !angle || !out || n < 1) return DV_ERR_INVALID;
    if (!c->have_sensor) return fail(c, DV_ERR_STATE, "sensor not configured");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t bytes = (size_t)n * c->sensor.sh * c->sensor.sw * 3;
    int rc = ensure_sense_buffer(c, bytes);
    if (rc) return rc;
    rc = enqueue_sense(c, x, y, angle, n, c->d_sense);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(out, c->d_sense, bytes, hipMemcpyDeviceToHost, c->stream));
    return check_sense_error(c);
}

static int shape_now(dv_ctx* c);
// The byte kernels (k_sad_tiles, k_sad_packed, k_sad_generic) read the patches as dwords of d_prep; a step on the matrix cores does
// not (the mixed layout's saturation pass does).
static bool needs_prep_dwords(dv_ctx* c) {
    return c->cfg.generic || c->mixed || !c->bits_ok || shape_now(c) != 6;
}
// ... and if a byte kernel is about to run on patches prepared without them: from the raw bytes, nothing else touched.
static int ensure_prep_dwords(dv_ctx* c) {
    if (c->prep_dwords_ready) return DV_OK;
    const dim3 grid((unsigned)(c->A * ((c->cfg.P + 255) / 256))), block(256);
    static const PoseSet no_poses{};
    hipLaunchKernelGGL(k_patch_prep<0>, grid, block, 0, c->stream, (const unsigned char*)nullptr, no_poses, c->A, SensorCfg{}, (const unsigned char*)nullptr,
                       c->d_raw_patches, c->d_prep, c->cfg, c->APAD, c->d_acc + c->acc_parity, c->d_acc + (c->acc_parity ^ 1), c->A_agent, PrepBits{},
                       0ull, (uint4*)nullptr, (uint4*)nullptr, 1, PathErrArgs{});
    HIP_TRY(c, hipGetLastError());
    c->prep_dwords_ready = true;
    return DV_OK;
}

// Senses the patches of A_total headings (poses by value) straight into the scoring kernel's operand layout: ONE
// kernel, no copy and no memset on the way (see k_sense_prep).  n_agents agents of A_agent headings each.
static int launch_patch_prep(dv_ctx* c, int mode, const PoseSet* poses, int n_agents, int A_agent, unsigned long long seed,
                             const PathErrArgs* path_err = nullptr) {
    Range range(mode == 1 ? "dv:sense" : "dv:prep");
    const int A = n_agents * A_agent;
    c->A = A; c->n_agents = n_agents; c->A_agent = A_agent;
    c->APAD = A <= 8 ? 8 : (A <= 16 ? 16 : (A <= 32 ? 32 : 64));
    c->acc_parity ^= 1;
    PrepAcc* cur = c->d_acc + c->acc_parity;
    PrepAcc* nxt = c->d_acc + (c->acc_parity ^ 1);
    // (prep entries and image columns of the padded headings A..APAD-1 are left as they are: their sums are never read)
    const PathErrArgs pe = path_err ? *path_err : PathErrArgs{};
    const dim3 grid((unsigned)(A * ((c->cfg.P + 255) / 256) + pe.nblk)), block(256);      // one block per (heading, 256 pixels) [+ the metric blocks]
    static const PoseSet no_poses{};
    const PrepBits pb = c->bits_ok ? c->pbits : PrepBits{};
    uint4* i8 = c->bits_ok ? c->d_coef : nullptr;
    uint4* i4 = (c->bits_ok && c->fp4_ok) ? c->d_coef4 : nullptr;
    // the byte path's operand dwords only where a byte kernel will read them (launch_int_scoring writes them later if one does after all)
    const bool dwords = needs_prep_dwords(c);
    const int what = 2 | (dwords ? 1 : 0);
    c->prep_dwords_ready = dwords;
    if (mode == 1)
        hipLaunchKernelGGL(k_patch_prep<1>, grid, block, 0, c->stream, c->d_land, *poses, A, c->sensor, c->d_lut, c->d_raw_patches, c->d_prep,
                           c->cfg, c->APAD, cur, nxt, A_agent, pb, 0ull, i8, i4, what, pe);
    else if (mode == 2)
        hipLaunchKernelGGL(k_patch_prep<2>, grid, block, 0, c->stream, (const unsigned char*)nullptr, no_poses, A, SensorCfg{}, (const unsigned char*)nullptr,
                           c->d_raw_patches, c->d_prep, c->cfg, c->APAD, cur, nxt, A_agent, pb, seed, i8, i4, what, pe);
    else
        hipLaunchKernelGGL(k_patch_prep<0>, grid, block, 0, c->stream, (const unsigned char*)nullptr, no_poses, A, SensorCfg{}, (const unsigned char*)nullptr,
                           c->d_raw_patches, c->d_prep, c->cfg, c->APAD, cur, nxt, A_agent, pb, 0ull, i8, i4, what, pe);
    HIP_TRY(c, hipGetLastError());
    c->patches_sensed = mode == 1;     // no host synchronisation here: the step's result record carries the sensor's error flag
    return enqueue_bit_prep(c);
}
static int sense_prep_launch(dv_ctx* c, const PoseSet& poses, int n_agents, int A_agent, const PathErrArgs* path_err = nullptr) {
    return launch_patch_prep(c, 1, &poses, n_agents, A_agent, 0ull, path_err);
}

static int check_sense_args(dv_ctx* c, int A) {
    int rc = check_step_args(c, A);
    if (rc) return rc;
    if (!c->have_sensor) return fail(c, DV_ERR_STATE, "sensor not configured");
    if (c->sensor.sw != c->w || c->sensor.sh != c->h)
        return fail(c, DV_ERR_STATE, "sensor is %dx%d but the library holds %dx%d views", c->sensor.sw, c->sensor.sh, c->w, c->h);
    return DV_OK;
}

static inline Pose make_pose(double x, double y, double angle) {   // cos/sin from the host's libm, as the reference's come
    const double rot = -(0.5 * M_PI - angle);
    return Pose{x, y, std::cos(rot), std::sin(rot)};
}

extern "C" int dv_sense_patches(dv_ctx* c, double x, double y, const double* angles, int A) {
    int rc = check_sense_args(c, A);
    if (rc) return rc;
    if (!angles) return fail(c, DV_ERR_INVALID, "angles is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    PoseSet poses;
    for (int a = 0; a < A; ++a) poses.p[a] = make_pose(x, y, angles[a]);
    for (int a = A; a < kMaxHeadings; ++a) poses.p[a] = Pose{0., 0., 1., 0.};
    return sense_prep_launch(c, poses, 1, A);
}

// The per-step buffers (and the few scalars that describe the resident patches) a pass writes, as sets that can be swapped into the
// context's fields: set 0 is the context's own, set j > 0 lives in c->extra[j - 1] while it is not in use.
static void swap_set(dv_ctx* c, dv_ctx::PassSet& a) {
    std::swap(c->d_raw_patches, a.raw); std::swap(c->d_coef, a.coef); std::swap(c->d_coef4, a.coef4);
    std::swap(c->d_acc, a.acc); std::swap(c->acc_parity, a.parity); std::swap(c->d_bsum, a.bsum);
    std::swap(c->d_ctmp, a.ctmp); std::swap(c->d_state, a.state); std::swap(c->d_cand, a.cand); std::swap(c->d_cand_exact, a.cand_exact);
    std::swap(c->coef_ready, a.coef_ready); std::swap(c->prep_dwords_ready, a.prep_dwords_ready); std::swap(c->patches_sensed, a.patches_sensed);
    std::swap(c->A, a.A); std::swap(c->APAD, a.APAD); std::swap(c->n_agents, a.n_agents); std::swap(c->A_agent, a.A_agent);
    std::swap(c->fused_nb, a.fused_nb);
}
static void use_set(dv_ctx* c, int which) {
    if (which == c->cur_set) return;
    if (c->cur_set != 0) swap_set(c, c->extra[c->cur_set - 1]);       // home first
    if (which != 0) swap_set(c, c->extra[which - 1]);
    c->cur_set = which;
}

// At least `want` extra sets (<= kExtraSets), the second stream and the two events; returns how many sets there are (0: the passes run
// one after the other as before).  Allocates at the first ensemble call on a library that wants them, never inside a single step.
static int ensure_extra_sets(dv_ctx* c, int want) {
    if (c->chains_env < 2 || c->metric != 0 || !c->bits_ok || c->mixed || c->stream != c->own_stream || c->cur_set != 0) return 0;
    if (want > dv_ctx::kExtraSets) want = dv_ctx::kExtraSets;
    if (!c->batch_stream) {
        // a stream of ANOTHER priority: the runtime maps streams of one priority onto a few hardware queues round-robin, and two
        // streams that share a queue run their kernels one after the other (rocprofv3 showed both on queue 4); queues of different
        // priority classes are distinct
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        if (hipStreamCreateWithPriority(&c->batch_stream, hipStreamNonBlocking, greatest) != hipSuccess) { (void)hipGetLastError(); c->batch_stream = nullptr; return 0; }
        if (c->chains_env == 3 && hipStreamCreateWithPriority(&c->batch_stream3, hipStreamNonBlocking, least) != hipSuccess) { (void)hipGetLastError(); c->batch_stream3 = nullptr; }
    }
    if (!c->ev_a && hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); c->ev_a = nullptr; return 0; }
    if (!c->ev_b && hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); c->ev_b = nullptr; return 0; }
    const LibCfg& g = c->cfg;
    const size_t nkt = (size_t)(c->bcfg.NK[0] + c->bcfg.NK[1]);
    while (c->n_extra < want) {
        dv_ctx::PassSet& a = c->extra[c->n_extra];
        bool ok = hipMalloc(&a.raw, (size_t)kMaxHeadings * g.P * 3) == hipSuccess &&
                  hipMalloc(&a.coef, (size_t)2 * nkt * 8192) == hipSuccess &&
                  (!c->fp4_ok || hipMalloc(&a.coef4, (size_t)2 * nkt * 4096) == hipSuccess) &&
                  hipMalloc(&a.acc, 2 * sizeof(PrepAcc)) == hipSuccess &&
                  hipMalloc(&a.bsum, (size_t)std::max<long long>((g.F + 255) / 256, 256) * 2 * kMaxHeadings * sizeof(unsigned long long)) == hipSuccess &&
                  hipMalloc(&a.ctmp, (size_t)kMaxHeadings * kTmpCap * 2 * sizeof(unsigned long long)) == hipSuccess &&
                  hipMalloc(&a.state, kMaxHeadings * sizeof(StepState)) == hipSuccess &&
                  hipMalloc(&a.cand, (size_t)kMaxHeadings * kCandCap * sizeof(unsigned long long)) == hipSuccess &&
                  hipMalloc(&a.cand_exact, (size_t)kMaxHeadings * kCandCap * sizeof(double)) == hipSuccess;
        if (ok)          // as the first set is initialised (alloc_library, build_bit_planes): cleared constants and state, zeroed images
            ok = hipMemsetAsync(a.acc, 0, 2 * sizeof(PrepAcc), c->stream) == hipSuccess &&
                 hipMemsetAsync(a.state, 0, kMaxHeadings * sizeof(StepState), c->stream) == hipSuccess &&
                 hipMemsetAsync(a.coef, 0, (size_t)2 * nkt * 8192, c->stream) == hipSuccess &&
                 (!c->fp4_ok || hipMemsetAsync(a.coef4, 0, (size_t)2 * nkt * 4096, c->stream) == hipSuccess) &&
                 hipStreamSynchronize(c->stream) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            auto F = [](auto*& p) { if (p) { (void)hipFree(p); p = nullptr; } };
            F(a.raw); F(a.coef); F(a.coef4); F(a.acc); F(a.bsum); F(a.ctmp); F(a.state); F(a.cand); F(a.cand_exact);
            a = dv_ctx::PassSet{};
            break;
        }
        ++c->n_extra;
    }
    return c->n_extra;
}

// Ensemble passes: the passes of up to 64 agents each take a result record of their own (`result_slot`) and the host collects the
// records afterwards.  The first pass runs on its own (it shows whether this library's passes finish their scores themselves: a fused
// pass writes nothing shared).  If so, the others go in groups of up to kExtraSets passes, each with its own set of per-step buffers:
//   1. every preparation of the group, on the context's stream;
//   2. the scoring kernels, alternating between the context's stream and a second one of another priority (= two hardware queues):
//      a kernel is 256 persistent workgroups with one CU each, so the next pass's workgroups move in as the last pass's finish their
//      third round -- the fourth, which only 14 of 256 run, no longer holds the chip (and no small kernel sits between two passes);
//   3. the folds (one small workgroup per agent and pass), last.
// A pass whose agents need the exact resolver, overflowed their candidate list or sensed past the landscape is simply run again on
// its own through the synchronous path, which handles all of that.  `stage(first, n)` makes agents [first, first + n) the resident
// patches.
template <class Stage>
static int run_batch(dv_ctx* c, int n_agents, int A, uint32_t flags, dv_step_result* results, Stage stage, bool stage_uploads) {
    struct Pass { int first, n, seq, slot; };
    const int per_pass = kMaxHeadings / A;
    int rc = DV_OK;
    const int full_total = per_pass * A;                                // resident headings of a full pass
    const int full_apad = full_total <= 8 ? 8 : (full_total <= 16 ? 16 : (full_total <= 32 ? 32 : 64));
    bool first_pass_seen = false;
    auto sync_both = [&]() {
        (void)hipStreamSynchronize(c->stream);
        if (c->batch_stream) (void)hipStreamSynchronize(c->batch_stream);
        if (c->batch_stream3) (void)hipStreamSynchronize(c->batch_stream3);
    };
    for (int sb = 0; sb < n_agents && rc == DV_OK; sb += kMaxHeadings) {
        const int cnt = (n_agents - sb < kMaxHeadings) ? n_agents - sb : kMaxHeadings;
        std::vector<Pass> passes;
        std::vector<std::pair<int, int>> todo;                           // (first agent of the 64, agents) of every pass of this 64
        for (int first = 0; first < cnt; first += per_pass) todo.push_back({first, (cnt - first < per_pass) ? cnt - first : per_pass});
        size_t k = 0;
        auto classic = [&](const std::pair<int, int>& p) -> int {        // preparation, scoring, fold on the context's stream
            int r2 = stage(sb + p.first, p.second);
            if (r2) return r2;
            c->result_slot = p.first;
            r2 = enqueue_step(c, flags, false);
            passes.push_back(Pass{sb + p.first, p.second, c->seq, p.first});
            return r2;
        };
        if (!first_pass_seen && !batch_pass_fuses(c, full_apad) && k < todo.size()) {
            rc = classic(todo[k++]);               // (an untimed heading class: this pass times the kernel forms; the rest may then group)
            first_pass_seen = true;
        }
        while (rc == DV_OK && k < todo.size()) {
            // passes of the full size only (a shorter last pass may take another kernel form: on its own, afterwards)
            int left = 0;
            while (k + (size_t)left < todo.size() && todo[k + (size_t)left].second == per_pass) ++left;
            const int nset = (left > 1 && batch_pass_fuses(c, full_apad)) ? ensure_extra_sets(c, left) : 0;
            if (nset < 2) { rc = classic(todo[k++]); continue; }
            const int ng = left < nset ? left : nset;
            const int force = (flags & DV_STEP_FORCE_RESOLVE) ? 1 : 0;
            std::vector<int> seqs((size_t)ng, 0);
            // Pass j of the group lives on stream j & 1 from its preparation to its fold: two chains that share nothing but the
            // library -- no event between the streams.  A chain's layout (DEJAVU_CHAIN_ORDER, tools/runs/r4_chain.sh; 32 agents x 16
            // headings, 100 000 views of 64x64, ms per ensemble step sensed / uploaded):
            //   0  [its preparations][its scoring kernels][its folds]          0.867-0.881 / 1.057-1.082
            //   1  [preparation, scoring kernel] per pass, [its folds]         0.892-0.920 / 0.999-1.021
            //   2  [preparation, scoring kernel, fold] per pass                0.893-0.911 / 1.052-1.077
            // A small kernel between two scoring kernels stalls its chain while the other chain's kernel has the chip to itself, and the
            // round-filling overlap of the two kernels is lost for that time: sensed patches (nothing but a 10-us preparation per pass)
            // keep layout 0; uploaded patches take layout 1, where a pass's host-to-device copy runs beside the pass before it.
            const int nch = c->batch_stream3 ? 3 : 2;
            auto on = [&](int j) { use_set(c, j + 1); const int ch = j % nch; c->stream = ch == 0 ? c->own_stream : (ch == 1 ? c->batch_stream : c->batch_stream3); };
            const int order = c->chain_order_env >= 0 ? c->chain_order_env : (stage_uploads ? 1 : 0);
            if (order == 0)
                for (int j = 0; j < ng && rc == DV_OK; ++j) { on(j); rc = stage(sb + todo[k + j].first, todo[k + j].second); }
            c->defer_fold = order != 2;
            for (int j = 0; j < ng && rc == DV_OK; ++j) {
                on(j);
                if (order != 0) rc = stage(sb + todo[k + j].first, todo[k + j].second);   // a pass's preparation right in front of its scoring kernel
                if (rc) break;
                c->result_slot = todo[k + j].first;
                rc = enqueue_step(c, flags, false);
                if (rc == DV_OK && !c->epilogue_fused) rc = fail(c, DV_ERR_STATE, "internal: an ensemble pass of a fused library did not fuse");
                seqs[(size_t)j] = order != 2 ? c->deferred_seq : c->seq;
                if (order == 2) passes.push_back(Pass{sb + todo[k + j].first, todo[k + j].second, seqs[(size_t)j], todo[k + j].first});
            }
            c->defer_fold = false;
            // the folds, one per pass on the pass's own chain.  (ONE launch folding all passes of the group behind both chains' last
            // kernels was measured and gave nothing: 0.856-0.859 against 0.847-0.852 ms per ensemble step -- the two chains' folds
            // already run side by side, and the host collects the records in the order they arrive.)
            for (int j = 0; j < ng && rc == DV_OK && order != 2; ++j) {
                on(j);
                const int slot = todo[k + j].first;
                launch_fold(c, c->fused_nb, c->d_result + slot, c->d_record + (size_t)slot * (3 + 4 * kMaxHeadings), force, seqs[(size_t)j],
                            sense_err_ptr(c));
                if (hipGetLastError() != hipSuccess) rc = fail(c, DV_ERR_HIP, "k_fold launch failed");
                passes.push_back(Pass{sb + slot, todo[k + j].second, seqs[(size_t)j], slot});
            }
            c->stream = c->own_stream;
            use_set(c, 0);
            k += (size_t)ng;
        }
        c->stream = c->own_stream;
        use_set(c, 0);
        c->defer_fold = false;
        c->result_slot = 0;
        if (rc) { sync_both(); return rc; }
        bool polled = c->spin_wait != 0;
        for (const Pass& p : passes)
            if (polled && !spin_for_records(c, p.slot, p.n, A, p.seq)) polled = false;
        if (!polled) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            if (c->batch_stream) HIP_TRY(c, hipStreamSynchronize(c->batch_stream));
            if (c->batch_stream3) HIP_TRY(c, hipStreamSynchronize(c->batch_stream3));
        }
        std::vector<Pass> again;
        for (const Pass& p : passes) {
            unsigned bad = 0;
            for (int ag = 0; ag < p.n; ++ag) bad |= c->h_result[p.slot + ag].flags & (kResNeedsResolve | DV_RES_OVERFLOW);
            if (bad) { again.push_back(p); continue; }            // (an agent sensed past the landscape just carries its flag)
            for (int ag = 0; ag < p.n; ++ag) {
                memcpy(&results[p.first + ag], &c->h_result[p.slot + ag], sizeof(dv_step_result));
                for (int a = A; a < kMaxHeadings; ++a) {
                    results[p.first + ag].angle_fam[a] = 0.0; results[p.first + ag].angle_view[a] = -1;
                    results[p.first + ag].exact_fam[a] = 0.0; results[p.first + ag].exact_view[a] = -1;
                }
            }
        }
        for (const Pass& p : again) {
            rc = stage(p.first, p.n);
            if (rc) return rc;
            rc = enqueue_step(c, flags, false);
            if (rc) return rc;
            rc = finish_pass(c);
            if (rc) return rc;
            for (int ag = 0; ag < p.n; ++ag) copy_result(c, ag, &results[p.first + ag]);
        }
    }
    return rc;
}

// Ensemble form of dv_sense_step: n_agents agents, each at its own position with its own A headings, sensed and
// scored against the one resident library, 64/A agents per library pass; nothing but poses goes up.
extern "C" int dv_sense_step_batch(dv_ctx* c, const double* x, const double* y, const double* angles, int n_agents, int A,
                                   uint32_t flags, dv_step_result* results) {
    int rc = check_sense_args(c, A);
    if (rc) return rc;
    if (!x || !y || !angles || !results || n_agents < 1) return fail(c, DV_ERR_INVALID, "dv_sense_step_batch: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    return run_batch(c, n_agents, A, flags, results, [&](int first, int n) {
        PoseSet poses;
        for (int ag = 0; ag < n; ++ag)
            for (int a = 0; a < A; ++a)
                poses.p[ag * A + a] = make_pose(x[first + ag], y[first + ag], angles[(size_t)(first + ag) * A + a]);
        for (int a = n * A; a < kMaxHeadings; ++a) poses.p[a] = Pose{0., 0., 1., 0.};
        return sense_prep_launch(c, poses, n, A);
    }, false);
}

// One full agent step's device work in one call: sense the heading patches at (x, y), score them, decide.
extern "C" int dv_sense_step(dv_ctx* c, double x, double y, const double* angles, int A, uint32_t flags,
                             dv_step_result* result, double* scene_fam) {
    int rc = dv_sense_patches(c, x, y, angles, A);
    if (rc) return rc;
    if (!result) return fail(c, DV_ERR_INVALID, "result is NULL");
    rc = enqueue_step(c, flags, scene_fam != nullptr);
    if (rc) return rc;
    return wait_step(c, result, scene_fam);
}

constexpr int kErrRing = 8;             // answers of the error metrics that may be outstanding
// Arguments of one error-metric computation; takes the next sequence number of the answer ring.
static PathErrArgs path_err_args(dv_ctx* c, double x, double y, double reach) {
    PathErrArgs pe{};
    const unsigned long long seq = ++c->err_enq;
    // blocks of 256 threads, 1024 points each: every block ends in two agent-scope atomics on ONE address (minimum, ticket), which the
    // memory side takes one after the other -- 196 blocks for a 50 000-point path were the longest thing in the preparation launch
    long long nb = (c->n_path + 1023) / 1024;
    if (nb > 256) nb = 256;
    pe.xy = c->d_path; pe.n = (long long)c->n_path; pe.x = x; pe.y = y; pe.reach = reach; pe.cover = c->d_cover; pe.st = c->d_errstate;
    pe.out = c->d_errout + (seq % kErrRing); pe.seq = seq; pe.nblk = (int)nb;
    return pe;
}

// One agent step's device work and device-side book-keeping (include/dejavu.h: dv_agent_step = dv_agent_step_begin + _end).
// begin: collect the error answer asked for a step ago, launch sensing / preparation (with the metric blocks of (ex, ey)) / scoring /
// fold, return; end: wait for the record, resolve near-ties, hand out the per-heading maxima and the decision.  Between the two the
// host is free: an agent calls begin for the step it will take next as soon as its new pose is known and does its book-keeping
// while the device works (navsim_amd/agent.py).  A begin that no end follows is harmless: later work queues behind it on the stream
// and every record carries its step's sequence number.
extern "C" int dv_agent_step_begin(dv_ctx* c, double x, double y, double angle, const double* offsets, int A, int do_error, double ex,
                                   double ey, double reach, double* nearest, int32_t* have_nearest) {
    int rc = check_sense_args(c, A);
    if (rc) return rc;
    if (!offsets || !nearest || !have_nearest) return fail(c, DV_ERR_INVALID, "dv_agent_step_begin: NULL argument");
    *have_nearest = 0;
    c->agent_pending = 0;
    if (do_error && c->n_path < 1) return fail(c, DV_ERR_STATE, "no training path set (dv_set_training_path)");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->err_enq > c->err_deq) {                                     // asked for a step ago: there by now
        rc = dv_path_error_wait(c, nearest);
        if (rc) return rc;
        *have_nearest = 1;
    }
    const double two_pi = 2.0 * M_PI;
    PoseSet poses;
    for (int a = 0; a < A; ++a) {
        double m = std::fmod(angle + offsets[a], two_pi);            // np.mod: a remainder of the divisor's sign
        if (m != 0.0) { if (m < 0.0) m += two_pi; } else m = 0.0;
        poses.p[a] = make_pose(x, y, m);
    }
    for (int a = A; a < kMaxHeadings; ++a) poses.p[a] = Pose{0., 0., 1., 0.};
    // the metrics of (ex, ey) ride in the preparation launch: blocks behind its own (no launch, no second stream)
    PathErrArgs pe{};
    if (do_error) {
        if (c->err_enq - c->err_deq >= (unsigned long long)kErrRing)
            return fail(c, DV_ERR_STATE, "%d path-error answers outstanding: collect them with dv_path_error_wait", kErrRing);
        if (!c->err_on_main) { HIP_TRY(c, hipStreamSynchronize(c->aux_stream)); c->err_on_main = true; }   // (shared d_errstate)
        pe = path_err_args(c, ex, ey, reach);
    }
    rc = sense_prep_launch(c, poses, 1, A, do_error ? &pe : nullptr);
    if (rc) return rc;
    rc = enqueue_step(c, 0, false);
    if (rc) return rc;
    c->agent_pending = A;
    return DV_OK;
}

extern "C" int dv_agent_step_end(dv_ctx* c, double* angle_fam, int32_t* best_heading) {
    if (!c) return DV_ERR_INVALID;
    if (!angle_fam || !best_heading) return fail(c, DV_ERR_INVALID, "dv_agent_step_end: NULL argument");
    if (!c->agent_pending || !c->step_pending) return fail(c, DV_ERR_STATE, "dv_agent_step_end: no agent step was begun (or another step came between)");
    const int A = c->agent_pending;
    c->agent_pending = 0;
    HIP_TRY(c, hipSetDevice(c->device));
    int rc = finish_pass(c);
    if (rc) return rc;
    if (c->h_result[0].flags & kResSenseError)
        return fail(c, DV_ERR_INDEX, "sensor footprint reaches past the end of the landscape (index out of bounds)");
    memcpy(angle_fam, c->h_result[0].angle_fam, (size_t)A * sizeof(double));
    *best_heading = c->h_result[0].best_heading;
    return DV_OK;
}

// The end of one agent step and the beginning of the next in ONE call: the record is waited for, and the step for the pose the chosen
// heading leads to -- cand_x / cand_y / cand_angle[best], worked out by the caller for every heading while the device was busy -- is
// begun at once, without a trip through the caller in between (include/dejavu.h).
extern "C" int dv_agent_step_end_begin(dv_ctx* c, double* angle_fam, int32_t* best_heading, const double* cand_x, const double* cand_y,
                                       const double* cand_angle, const double* offsets, int A, const double* bounds, int do_error, double reach,
                                       int32_t* begun, double* nearest, int32_t* have_nearest) {
    if (!c) return DV_ERR_INVALID;
    if (!cand_x || !cand_y || !cand_angle || !bounds || !begun || !nearest || !have_nearest)
        return fail(c, DV_ERR_INVALID, "dv_agent_step_end_begin: NULL argument");
    *begun = 0;
    *have_nearest = 0;
    if (c->agent_pending && c->agent_pending != A) return fail(c, DV_ERR_INVALID, "dv_agent_step_end_begin: %d headings begun, %d asked for", c->agent_pending, A);
    int rc = dv_agent_step_end(c, angle_fam, best_heading);
    if (rc) return rc;
    const int b = *best_heading;
    if (b < 0 || b >= A) return fail(c, DV_ERR_STATE, "dv_agent_step_end_begin: heading %d of %d", b, A);
    const double x = cand_x[b], y = cand_y[b];
    // the reference's bounds test (NavBySceneFamiliarity.py:153-158): out of bounds, the next step stops before it senses -- nothing is begun
    if ((x <= bounds[0]) || (y <= bounds[0]) || (x >= bounds[1]) || (y >= bounds[2])) return DV_OK;
    rc = dv_agent_step_begin(c, x, y, cand_angle[b], offsets, A, do_error, x, y, reach, nearest, have_nearest);
    if (rc) return rc;
    *begun = 1;
    return DV_OK;
}

extern "C" int dv_agent_step(dv_ctx* c, double x, double y, double angle, const double* offsets, int A, int do_error, double ex,
                             double ey, double reach, double* angle_fam, int32_t* best_heading, double* nearest, int32_t* have_nearest) {
    if (c && (!angle_fam || !best_heading)) return fail(c, DV_ERR_INVALID, "dv_agent_step: NULL argument");
    int rc = dv_agent_step_begin(c, x, y, angle, offsets, A, do_error, ex, ey, reach, nearest, have_nearest);
    if (rc) return rc;
    return dv_agent_step_end(c, angle_fam, best_heading);
}

extern "C" int dv_set_library_from_poses(dv_ctx* c, const double* x, const double* y, const double* angle, int64_t n,
                                         double cw, int64_t first, uint8_t* out_views) {
    if (!c || !x || !y || !angle) return DV_ERR_INVALID;
    if (!c->have_sensor) return fail(c, DV_ERR_STATE, "sensor not configured");
    int rc = check_lib_args(c, n, c->sensor.sh, c->sensor.sw, cw);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_library(c);
    const size_t bytes = (size_t)n * c->sensor.sh * c->sensor.sw * 3;
    rc = ensure_sense_buffer(c, bytes);
    if (rc) return rc;
    rc = enqueue_sense(c, x, y, angle, n, c->d_sense);
    if (rc) return rc;
    if (out_views) HIP_TRY(c, hipMemcpyAsync(out_views, c->d_sense, bytes, hipMemcpyDeviceToHost, c->stream));
    rc = check_sense_error(c);
    if (rc) return rc;
    return ingest_raw(c, c->d_sense, n, c->sensor.sh, c->sensor.sw, cw, first);
}

extern "C" int dv_generate_library(dv_ctx* c, uint64_t seed, int64_t F, int h, int w, double cw, int64_t first) {
    return dv_generate_library_ex(c, seed, F, h, w, cw, first, 0);
}

extern "C" int dv_generate_library_ex(dv_ctx* c, uint64_t seed, int64_t F, int h, int w, double cw, int64_t first, int full_range_s) {
    int rc = check_lib_args(c, F, h, w, cw);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    const unsigned char hues[2] = {0, 127};   // synth.hsv_from_words: H = bit * 127, S > 0 in both
    rc = alloc_library(c, F, h, w, cw, first, 2, hues, 0, 127);     // synth: S is 0 or 127
    if (rc) { free_library(c); return rc; }
    c->cfg.synth_full_s = full_range_s ? 1 : 0;
    const long long total = (c->cfg.Fpad / 64) * (long long)c->cfg.npl * c->cfg.Q * 64;
    hipLaunchKernelGGL(k_generate_tiles, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->d_tiles,
                       c->cfg, (unsigned long long)seed);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    rc = build_bit_planes(c);
    if (rc) { free_library(c); return rc; }
    return DV_OK;
}

extern "C" int dv_patches_on_level(dv_ctx* c) {
    if (!c) return DV_ERR_INVALID;
    if (!c->have_lib || !c->bits_ok || !c->fp4_ok) return fail(c, DV_ERR_STATE, "this library has no fp4 form");
    if (!c->coef_ready) return fail(c, DV_ERR_STATE, "no coefficient image of the resident patches yet");
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned word = 0;
    HIP_TRY(c, hipMemcpyAsync(&word, offlevel_word(c), sizeof word, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return word == 0 ? 1 : 0;
}

extern "C" int dv_scoring_form(dv_ctx* c) {
    if (!c) return DV_ERR_INVALID;
    int form = c->last_form;
    if (form & DV_FORM_FP4) {                              // the dual kernel ran: which image it took is on the device
        HIP_TRY(c, hipSetDevice(c->device));
        unsigned word = 0;
        HIP_TRY(c, hipMemcpyAsync(&word, offlevel_word(c), sizeof word, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (word != 0) form &= ~DV_FORM_FP4;
    }
    return form;
}

extern "C" int dv_clear_library(dv_ctx* c) {
    if (!c) return DV_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    free_library(c);
    return DV_OK;
}

extern "C" int dv_get_library_info(const dv_ctx* c, dv_lib_info* o) {
    if (!c || !o) return DV_ERR_INVALID;
    if (!c->have_lib) return DV_ERR_STATE;
    memset(o, 0, sizeof *o);
    o->n_views = c->cfg.F;
    o->first_view = c->cfg.first;
    o->h = c->h;
    o->w = c->w;
    o->n_planes = c->cfg.npl;
    o->n_hue_planes = c->cfg.generic ? 0 : c->cfg.nhs;
    o->generic_hue = c->cfg.generic;
    o->signed_saturation = c->cfg.signed_s;
    o->has_value_plane = c->cfg.hasv;
    o->tile_bytes = (int64_t)c->tile_bytes;
    o->chem_weight = c->cfg.cw;
    o->delta = c->delta;
    for (int k = 0; k < kMaxHues; ++k) o->hues[k] = c->cfg.hues[k];
    o->n_hues = 0;
    if (c->cfg.cw > 0.0 && !c->cfg.generic) o->n_hues = c->cfg.signed_s ? 2 : c->cfg.nhs;
    o->has_bit_planes = c->bits_ok ? 1 : 0;
    o->fp4_form = (c->bits_ok && c->fp4_ok) ? 1 : 0;
    o->bit_planes_hs = c->bits_ok ? c->bcfg.T[0] : 0;
    o->bit_planes_v = c->bits_ok ? c->bcfg.T[1] : 0;
    o->bit_tile_bytes = c->bits_ok ? (int64_t)c->btile_bytes : 0;
    o->code_tile_bytes = (c->bits_ok && c->bcfg.vcode) ? (int64_t)c->ctile_bytes : 0;
    o->mixed_layout = (c->bits_ok && c->mixed) ? 1 : 0;
    return DV_OK;
}

extern "C" int dv_read_planes(dv_ctx* c, int64_t v0, int64_t n, uint8_t* out) {
    if (!c || !out) return DV_ERR_INVALID;
    if (!c->have_lib) return fail(c, DV_ERR_STATE, "no library set");
    if (v0 < 0 || n < 1 || v0 + n > c->cfg.F) return fail(c, DV_ERR_INVALID, "view range [%lld, %lld) outside the library", (long long)v0, (long long)(v0 + n));
    HIP_TRY(c, hipSetDevice(c->device));
    const long long total = n * c->cfg.npl * (long long)c->cfg.P;
    unsigned char* d = nullptr;
    HIP_TRY(c, hipMalloc(&d, (size_t)total));
    hipLaunchKernelGGL(k_read_planes, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, c->stream, c->d_tiles, d, c->cfg,
                       (long long)v0, (long long)n);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out, d, (size_t)total, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(c, DV_ERR_HIP, "read_planes: %s", hipGetErrorString(e));
    return DV_OK;
}

// ------------------------------------------------------------------ patches
static int prep_patches(dv_ctx* c, int A) { return launch_patch_prep(c, 0, nullptr, 1, A, 0ull); }

static int check_step_args(dv_ctx* c, int A) {
    if (!c) return DV_ERR_INVALID;
    if (!c->have_lib) return fail(c, DV_ERR_STATE, "no library set (call dv_set_library first)");
    if (c->metric != 0) return fail(c, DV_ERR_STATE, "the resident library is ssd_f32 / ssd_u8; use the _f32 / _u8 entry points");
    if (A < 1 || A > kMaxHeadings)
        return fail(c, DV_ERR_INVALID, "n_headings %d outside [1, %d] (dv_step_wide / dv_sense_step_wide take more)", A, kMaxHeadings);
    return DV_OK;
}

extern "C" int dv_upload_patches(dv_ctx* c, const uint8_t* patches, int A) {
    int rc = check_step_args(c, A);
    if (rc) return rc;
    if (!patches) return fail(c, DV_ERR_INVALID, "patches is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_raw_patches, patches, (size_t)A * c->cfg.P * 3, hipMemcpyHostToDevice, c->stream));
    return prep_patches(c, A);
}

extern "C" int dv_generate_patches(dv_ctx* c, uint64_t seed, int A) {
    int rc = check_step_args(c, A);
    if (rc) return rc;
    HIP_TRY(c, hipSetDevice(c->device));
    return launch_patch_prep(c, 2, nullptr, 1, A, (unsigned long long)seed);     // generated straight into the operand layouts
}

// ------------------------------------------------------------------ scoring launches
// Resident waves per CU of a scoring kernel: its VGPR allocation (granule 8, 512 per SIMD lane) and at most 7 per
// SIMD -- the kernels are built with <= 96 SGPRs (amdgpu_num_sgpr), which the hardware admits 7 times per SIMD.
static int resident_waves_per_cu(const void* kernel) {
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, kernel) != hipSuccess || attr.numRegs <= 0) return 8;
    const int alloc = (attr.numRegs + 7) / 8 * 8;
    int per_simd = 512 / alloc;
    if (per_simd > 7) per_simd = 7;
    if (per_simd < 1) per_simd = 1;
    return 4 * per_simd;
}

static bool kernel_uses_scratch(const void* kernel) {
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, kernel) != hipSuccess) { (void)hipGetLastError(); return false; }
    return attr.localSizeBytes > 0;
}

// Grid of the scoring kernels: workgroups of `wpb` waves, never more than are resident at once, walking the items
// with a grid stride.  The pixel range is cut into as many chunks as make the item count just fill the resident
// workgroups (measured optimum for single-wave workgroups: 7038 items on 7168 wave slots; one item more than fits costs
// a second round, fewer items leave SIMDs short of waves to hide latency).
static dim3 scoring_grid(dv_ctx* c, int kernel_wpc, dim3& block, int wpb = 1, size_t lds_bytes = 0) {
    const int wpc = c->waves_per_cu ? c->waves_per_cu : kernel_wpc;
    const long long G = c->cfg.Fpad / 64;
    long long per_cu = wpc / wpb > 0 ? wpc / wpb : 1;
    if (lds_bytes && per_cu > (long long)(163840 / lds_bytes)) per_cu = 163840 / lds_bytes;      // 160 KB of LDS per CU
    if (per_cu < 1) per_cu = 1;
    const long long slots = 256ll * per_cu;                                // workgroups resident at once
    long long n = c->target_items ? (c->target_items + G - 1) / G : slots / G;
    if (n < 1) n = 1;
    if (n > c->nchunk_cap) n = c->nchunk_cap;
    if (c->hs_bytes_pass) n = 1;                                             // its sums share the matrix-core pass's one-chunk rows
    c->nchunk = (int)n;
    const long long n_items = G * n;
    block = dim3(64 * wpb);
    return dim3((unsigned)(slots < n_items ? slots : n_items));
}

// One workgroup shape of k_sad_tiles: AP headings per wave, NW waves sharing an item's pixels, HW heading ways.
template <int NHS, int HASV, int AP, int ATOT, int NW, int HW>
static void launch_tiles_shape(dv_ctx* c) {
    static const int wpc = resident_waves_per_cu((const void*)k_sad_tiles<NHS, HASV, AP, ATOT, NW, HW>);
    dim3 block;
    constexpr int nsum = (NHS > 0 ? 1 : 0) + HASV;
    const size_t lds = NW > 1 ? (size_t)HW * nsum * AP * 64 * sizeof(unsigned) : 0;
    const dim3 grid = scoring_grid(c, wpc, block, NW * HW, lds);
    // more resident headings than one launch covers: further passes over the library
    for (int a_off = 0; a_off < ATOT; a_off += AP * HW)
        hipLaunchKernelGGL((k_sad_tiles<NHS, HASV, AP, ATOT, NW, HW>), grid, block, lds, c->stream, c->d_tiles, c->d_prep,
                           c->d_part, c->cfg, c->nchunk, a_off);
}

// The packed kernel (both sums in one accumulator per heading): all resident headings in one pass.
template <int NHS, int AP, int PF>
static void launch_packed(dv_ctx* c) {
    static const int wpc = resident_waves_per_cu((const void*)k_sad_packed<NHS, AP, PF>);
    dim3 block;
    const size_t lds = (size_t)2 * AP * 64 * sizeof(unsigned);
    const dim3 grid = scoring_grid(c, wpc, block, 4, lds);
    hipLaunchKernelGGL((k_sad_packed<NHS, AP, PF>), grid, block, lds, c->stream, c->d_tiles, c->d_prep, c->d_part, c->cfg,
                       c->nchunk);
}

template <int NHS>
static void launch_packed_apad(dv_ctx* c) {
    if (c->APAD == 8) launch_packed<NHS, 8, 1>(c);
    else if (c->APAD == 16) launch_packed<NHS, 16, 1>(c);
    else if (c->APAD == 32) launch_packed<NHS, 32, 1>(c);
    else launch_packed<NHS, 64, 1>(c);
}

// Workgroup shapes of k_sad_tiles per heading class (resident headings padded to 8, 16, 32 or 64):
//   1: single-wave workgroups, the pixel range cut finest (best CU balance);
//   2: four waves share an item and add their sums in LDS -- a quarter of the partial sums cross HBM (written here,
//      read again by k_combine) and the waves read the same patch dwords together;
//   3, 4: heading ways -- 2 or 4 waves score the same tiles against different slices of the headings, so the
//      library crosses HBM once for all of them and each wave keeps the register footprint (hence the occupancy)
//      of the narrower kernel; with 64 headings shapes 1 and 2 take two passes of the 32-wide kernel instead.
// Which one wins depends on the library size and the heading count (tools/sweep_grid.sh), so the shapes are timed once
// per library and heading class (tune_workgroup_shape).  The integer sums are identical in every shape.
static int apad_class(int APAD) { return APAD == 8 ? 0 : (APAD == 16 ? 1 : (APAD == 32 ? 2 : 3)); }
//   5: k_sad_packed (libraries with both sums only), all resident headings in one pass;
//   6: k_sad_mfma on the bit-plane copy of the library (libraries whose values come from few levels).
constexpr int kMaxShape = 6;
static bool shape_valid(const dv_ctx* c, int cls, int sh) {
    if (sh >= 1 && sh <= (cls == 0 ? 2 : (cls == 3 ? 4 : 3))) return true;
    if (sh == 6) return c->bits_ok;
    const bool both_sums = !c->cfg.generic && c->cfg.nhs > 0 && c->cfg.hasv;
    return sh == 5 && both_sums;
}
static int shape_now(dv_ctx* c) {
    const int cls = apad_class(c->APAD);
    int s = c->force_shape ? c->force_shape : (c->shape_env ? c->shape_env : c->tuned_shape[cls]);
    if (!shape_valid(c, cls, s)) s = 1;
    return s;
}

template <int NHS, int HASV>
static void launch_tiles_apad(dv_ctx* c) {
    const int s = shape_now(c);
    if constexpr (NHS > 0 && HASV == 1) {
        if (s == 5) return launch_packed_apad<NHS>(c);
    }
    if (c->APAD == 8) {
        if (s == 2) launch_tiles_shape<NHS, HASV, 8, 8, 4, 1>(c);
        else launch_tiles_shape<NHS, HASV, 8, 8, 1, 1>(c);
    } else if (c->APAD == 16) {
        if (s == 2) launch_tiles_shape<NHS, HASV, 16, 16, 4, 1>(c);
        else if (s == 3) launch_tiles_shape<NHS, HASV, 8, 16, 1, 2>(c);
        else launch_tiles_shape<NHS, HASV, 16, 16, 1, 1>(c);
    } else if (c->APAD == 32) {
        if (s == 2) launch_tiles_shape<NHS, HASV, 32, 32, 4, 1>(c);
        else if (s == 3) launch_tiles_shape<NHS, HASV, 16, 32, 1, 2>(c);
        else launch_tiles_shape<NHS, HASV, 32, 32, 1, 1>(c);
    } else {
        if (s == 2) launch_tiles_shape<NHS, HASV, 32, 64, 4, 1>(c);
        else if (s == 3) launch_tiles_shape<NHS, HASV, 32, 64, 1, 2>(c);
        else if (s == 4) launch_tiles_shape<NHS, HASV, 16, 64, 1, 4>(c);
        else launch_tiles_shape<NHS, HASV, 32, 64, 1, 1>(c);
    }
}

template <int HAS_HS, int HASV, int AP, int ATOT>
static void launch_generic(dv_ctx* c) {
    static const int wpc = resident_waves_per_cu((const void*)k_sad_generic<HAS_HS, HASV, AP, ATOT>);
    dim3 block;
    const dim3 grid = scoring_grid(c, wpc, block);
    for (int a_off = 0; a_off < ATOT; a_off += AP)
        hipLaunchKernelGGL((k_sad_generic<HAS_HS, HASV, AP, ATOT>), grid, block, 0, c->stream, c->d_tiles, c->d_prep, c->d_part,
                           c->cfg, c->nchunk, a_off);
}

template <int HAS_HS, int HASV>
static void launch_generic_apad(dv_ctx* c) {
    if (c->APAD == 8) launch_generic<HAS_HS, HASV, 8, 8>(c);
    else if (c->APAD == 16) launch_generic<HAS_HS, HASV, 16, 16>(c);
    else if (c->APAD == 32) launch_generic<HAS_HS, HASV, 32, 32>(c);
    else launch_generic<HAS_HS, HASV, 32, 64>(c);
}

static FuseArgs fuse_args(const dv_ctx* c) {
    FuseArgs fz{};
    fz.hsconst = c->d_acc[c->acc_parity].bhs;
    fz.vconst = c->d_acc[c->acc_parity].bv;
    fz.bsum = c->d_bsum;
    fz.ctmp = c->d_ctmp;
    fz.st = c->d_state;
    fz.A_real = c->A;
    fz.A_agent = c->A_agent;
    fz.delta = c->delta;
    return fz;
}

// View-group ranges (items per chunk) the library is cut into for a workgroup of 8 waves x TILES groups: as few as hold it,
// ceil(G32 / VW), of equal size to within one group -- 500 000 views x 128x128: 7.63 ranges of 8 per workgroup, so 94 of the 256
// workgroups sit out the eighth round.  Two finer cuts were measured and are SLOWER, because a range takes its consumers' time per
// stage whatever the bytes it streams (tools/exp/stamps.py):
//   round 2: every range finer (one per CU for small libraries, whole rounds of the 256 workgroups for large ones): 50 000 views x
//     64x64 ring loop 28.7 -> 30.3 us, 500 000 x 128x128 1.026 -> 1.035 ms;
//   round 3: whole ranges of 8 for the whole rounds and only the last round's cut evenly over all 256 workgroups: every workgroup
//     then ends within 50 us of the others, but later -- 500 000 views 0.969 -> 0.977 ms, 50 000 views loop 26.8 -> 28.1 us.
static long long item_groups(long long G32, int VW) {
    return (G32 + VW - 1) / VW;
}

// Both forms in one launch (k_sad_mfma_dual): the fp4 form when this prep's patches sit on the library's levels (the
// device decides, offlevel_word); libraries without an fp4 form point that word at a constant 1 and pass no fp4 image.
template <int SK8, int RD8, int SK4, int RD4, int TILES, bool FUSE, int SKL, int RDL, bool LCODE, int HT>
static void launch_mfma_dual_f(dv_ctx* c, int nchunk, int has_hs) {
    static bool attr_set = false;
    const size_t lds8 = (size_t)RD8 * (SK8 * 8 + 8 * SK8 * TILES) * 1024;
    const size_t lds4 = (size_t)fp4_ring_bytes(SK4, TILES, RD4);
    size_t lds = lds8 > lds4 ? lds8 : lds4;
    if constexpr (SKL > 0) {
        constexpr int lcb = lc_ring_bytes<SKL, RDL>();
        static_assert(lcb + kFuseScratchBytes + (HT - 1) * 512 <= 160 * 1024, "LDS");
        if ((size_t)lcb > lds) lds = (size_t)lcb;
    }
    lds += FUSE ? (size_t)kFuseScratchBytes + (HT - 1) * 512 : 0;     // (a second heading tile adds its 64 running-summary words)
    static_assert(fp4_ring_bytes(SK4, TILES, RD4) + kFuseScratchBytes <= 160 * 1024, "LDS");
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_sad_mfma_dual<SK8, RD8, SK4, RD4, TILES, FUSE, SKL, RDL, LCODE, HT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const long long G32 = c->cfg.Fpad / 32;
    const long long n_gq = item_groups(G32, 8 * TILES / HT);
    const long long items = n_gq * nchunk;
    const unsigned grid = (unsigned)(items < 256 ? items : 256);          // one 8-wave workgroup per CU, grid-stride
    const int nkt = c->bcfg.NK[0] + c->bcfg.NK[1];
    FuseArgs fz{};
    if (FUSE) { fz = fuse_args(c); fz.nb = (int)grid; }
    for (int a_off = 0; a_off < c->APAD; a_off += 32 * HT)
        hipLaunchKernelGGL((k_sad_mfma_dual<SK8, RD8, SK4, RD4, TILES, FUSE, SKL, RDL, LCODE, HT>), dim3(grid), dim3(512), lds, c->stream, c->d_btiles,
                           c->bcfg.vcode ? c->d_ctiles : c->d_btiles, c->d_coef + (size_t)(a_off / 32) * nkt * 512,
                           c->fp4_ok ? c->d_coef4 + (size_t)(a_off / 32) * nkt * 256 : nullptr, offlevel_word(c), reinterpret_cast<int*>(c->d_part),
                           c->cfg, c->bcfg, nchunk, c->APAD, a_off, has_hs, fz, (int)n_gq);
    if (FUSE) { c->epilogue_fused = true; c->fused_nb = (int)grid; }     // one summary per workgroup
}

template <int SK8, int RD8, int SK4, int RD4, int TILES, int SKL = 0, int RDL = 3, bool LCODE = false, int HT = 1>
static void launch_mfma_dual(dv_ctx* c, int nchunk, int has_hs) {
    // One chunk and a step that may end in k_fold: the kernel finishes its scores itself.
    if (c->fuse_request && nchunk == 1 && c->fuse_env) launch_mfma_dual_f<SK8, RD8, SK4, RD4, TILES, true, SKL, RDL, LCODE, HT>(c, nchunk, has_hs);
    else launch_mfma_dual_f<SK8, RD8, SK4, RD4, TILES, false, SKL, RDL, LCODE, HT>(c, nchunk, has_hs);
}

// Work items of k_sad_mfma_dual = (chunk of K-steps, range of at most 8*TILES view groups of 32).  Two view groups per wave
// halve the coefficient traffic (every A operand serves both) once the library is large enough to keep every CU busy that
// way; very small libraries also cut the K-steps into chunks so that there are about as many items as CUs.
struct MfmaPlan { bool use_lc, two_tiles, lc22; int tiles, nchunk; };
// k_sad_lc22 (two view groups x two heading tiles per consumer, bit positions of one width sharing an accumulator) fits this library:
// per segment the positions 1, 2, 3 that stand for something have one width, the saturation segment has one width altogether and
// its counts fit the int16 they wait in (sad_lc22_fp4).
static bool lc22_fits(const dv_ctx* c) {
    if (!c->lc22_env || !c->fp4_ok || c->bcfg.vcode || c->mixed) return false;
    for (int seg = 0; seg < 2; ++seg) {
        int w = 0;
        for (int bit = 1; bit < 4; ++bit) {
            const int wb = c->bcfg.wacc[seg][bit];
            if (wb && w && wb != w) return false;
            if (wb) w = wb;
        }
        if (seg == 0 && c->bcfg.wacc[0][0] && w && c->bcfg.wacc[0][0] != w) return false;
    }
    return (long long)c->bcfg.NK[0] * 256 <= 32767;
}
// How a matrix-core pass over the resident library is cut for `apad` resident headings (what launch_mfma launches; run_batch asks
// beforehand whether its passes will finish their scores themselves: one chunk).
static MfmaPlan mfma_plan(dv_ctx* c, int apad, bool fuse_request) {
    MfmaPlan p{};
    const long long G32 = c->cfg.Fpad / 32;
    // two view groups per wave once there are about 1.25 such items per CU (200 000 views x 128x128 x 32 headings, 391 items:
    // 0.432 ms with two, 0.474 ms with one; 500 000 views: two)
    // DEJAVU_LC (A/B): 0 = every wave loads and multiplies (sad_ring_fp4); 1 = loader and consumer waves, stage of 4 K-steps, ring of 3
    // (sad_lc_fp4; ranges of 8 view groups whatever the library's size); 2 = the same with stages of 2 K-steps, ring of 5
    p.use_lc = c->lc_env != 0 && c->fp4_ok && !c->mfma_tiles_env;
    p.tiles = c->mfma_tiles_env ? c->mfma_tiles_env : (p.use_lc ? 1 : (G32 >= 16ll * 320 ? 2 : 1));
    // (never by default a variant the compiler could only build with scratch: its fused form keeps a few item-level pointers
    // there in this build -- tests/test_host_logic.py:test_shipped_scoring_kernels_use_no_scratch lists what is guarded)
    if (p.tiles == 2 && !c->mfma_tiles_env) {
        static const bool spills[2] = {kernel_uses_scratch((const void*)k_sad_mfma_dual<1, 3, 2, 3, 2, false, 0, 3, false, 1>),
                                       kernel_uses_scratch((const void*)k_sad_mfma_dual<1, 3, 2, 3, 2, true, 0, 3, false, 1>)};
        const bool will_fuse = fuse_request && c->fuse_env && !c->mfma_chunk_env && item_groups(G32, 16) >= 160;
        if (spills[will_fuse ? 1 : 0]) p.tiles = 1;
    }
    // DEJAVU_HT=1 (A/B): 64 resident headings as two passes over the library instead of two heading tiles per view group in one
    p.two_tiles = p.use_lc && apad == 64 && c->ht_env == 2;
    p.lc22 = p.two_tiles && fuse_request && c->fuse_env && !c->mfma_chunk_env && lc22_fits(c) && item_groups(G32, 8) >= 160;
    const long long GQ = item_groups(G32, p.lc22 ? 8 : (p.two_tiles ? 4 : 8 * p.tiles));
    int nchunk = 1;
    if (c->mfma_chunk_env) {
        nchunk = c->mfma_chunk_env;
    } else if (GQ < 160) {
        // Fewer view groups than ~60 % of the CUs: cut the K-steps too.  (More chunks mean partial sums through HBM, k_finish
        // behind the kernel and more pipeline fills: only where CUs would otherwise have nothing at all.)
        nchunk = (int)((256 + GQ - 1) / GQ);
    }
    const int nk_min = c->bcfg.NK[1] > 0 ? (c->bcfg.NK[0] > 0 && c->bcfg.NK[0] < c->bcfg.NK[1] ? c->bcfg.NK[0] : c->bcfg.NK[1]) : c->bcfg.NK[0];
    while (nchunk > 1 && nk_min / nchunk < 4) --nchunk;                  // keep a few K-steps per chunk
    if (nchunk > c->nchunk_cap) nchunk = c->nchunk_cap;
    if (nchunk < 1 || c->mixed) nchunk = 1;                              // (mixed layout: the byte pass shares the one-chunk rows of the partial sums)
    p.nchunk = nchunk;
    if (nchunk != 1) p.lc22 = false;
    return p;
}

template <int SKL, int RDL>
static void launch_lc22(dv_ctx* c, int has_hs) {
    static bool attr_set = false;
    size_t lds = (size_t)lc22_ring_bytes<SKL, RDL>() + kLc22ParkBytes;
    const size_t lds8 = (size_t)2 * (4 * 8 + 8 * 4) * 1024;             // the int8 ring body's (off-level patches): <4, 1, 2>
    if (lds8 > lds) lds = lds8;
    lds += (size_t)kFuseScratchBytes + 512;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)k_sad_lc22<SKL, RDL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const long long G32 = c->cfg.Fpad / 32;
    const long long n_gq = item_groups(G32, 8);
    const unsigned grid = (unsigned)(n_gq < 256 ? n_gq : 256);          // one 8-wave workgroup per CU, grid-stride
    const int nkt = c->bcfg.NK[0] + c->bcfg.NK[1];
    FuseArgs fz = fuse_args(c);
    fz.nb = (int)grid;
    for (int a_off = 0; a_off < c->APAD; a_off += 64)
        hipLaunchKernelGGL((k_sad_lc22<SKL, RDL>), dim3(grid), dim3(512), lds, c->stream, c->d_btiles, c->d_coef + (size_t)(a_off / 32) * nkt * 512,
                           c->d_coef4 + (size_t)(a_off / 32) * nkt * 256, offlevel_word(c), c->cfg, c->bcfg, c->APAD, a_off, has_hs, fz, (int)n_gq);
    c->epilogue_fused = true;
    c->fused_nb = (int)grid;
}

// An ensemble pass of `apad` resident headings would finish its scores inside the scoring kernel (fused epilogue: nothing shared is
// written, so passes may run beside each other).  The shape must be known already (timed or forced): an untimed class says no.
static bool batch_pass_fuses(dv_ctx* c, int apad) {
    if (c->metric != 0 || c->exact || c->cfg.generic || !c->bits_ok || c->mixed || !c->fuse_env) return false;
    const int cls = apad_class(apad);
    const int shape = c->shape_env ? c->shape_env : c->tuned_shape[cls];
    if (shape != 6) return false;
    return mfma_plan(c, apad, true).nchunk == 1;
}

static void launch_mfma(dv_ctx* c, int has_hs) {
    const MfmaPlan plan = mfma_plan(c, c->APAD, c->fuse_request);
    const bool use_lc = plan.use_lc, two_tiles = plan.two_tiles;
    const int tiles = plan.tiles, nchunk = plan.nchunk, lc = c->lc_env;
    c->nchunk = nchunk;
    // <int8 stage, ring | fp4 stage, ring (thermometer rows) | fp4 stage, ring (code rows), view groups per wave>.  Measured in
    // round 2 (other ring shapes: DESIGN.md section 4): int8 500 000 views x 128x128 x 32 headings <1, 3> 1.29 ms, 50 000 views
    // x 64x64 x 16 headings <4, 2> 46.7 us; fp4 <2, 3> 0.95 ms and <2, 4> 34.5 us.
    const int ring = c->ring_env;                                                                   // A/B of ring shapes
    if (plan.lc22) launch_lc22<2, 3>(c, has_hs);
    else if (two_tiles && c->bcfg.vcode) launch_mfma_dual<4, 2, 2, 4, 1, 4, 3, true, 2>(c, nchunk, has_hs);
    else if (two_tiles) launch_mfma_dual<4, 2, 2, 4, 1, 4, 3, false, 2>(c, nchunk, has_hs);
    else if (use_lc && c->bcfg.vcode) launch_mfma_dual<4, 2, 2, 4, 1, 4, 3, true>(c, nchunk, has_hs);
    else if (use_lc && lc == 2) launch_mfma_dual<4, 2, 2, 4, 1, 2, 5>(c, nchunk, has_hs);
    else if (use_lc) launch_mfma_dual<4, 2, 2, 4, 1, 4, 3>(c, nchunk, has_hs);
    else if (tiles == 2) launch_mfma_dual<1, 3, 2, 3, 2>(c, nchunk, has_hs);
    else if (ring == 1) launch_mfma_dual<4, 2, 2, 6, 1>(c, nchunk, has_hs);
    else if (ring == 2) launch_mfma_dual<4, 2, 4, 3, 1>(c, nchunk, has_hs);
    else launch_mfma_dual<4, 2, 2, 4, 1>(c, nchunk, has_hs);
}

// The integer path of one scoring pass: k_sad_tiles / k_sad_generic, then k_combine.  `after_tiles` (optional) is
// recorded between the two.
static int launch_int_scoring(dv_ctx* c, hipEvent_t after_tiles, int* n_partial, bool with_combine = true) {
    const LibCfg& g = c->cfg;
    int has_hs_sum, has_v_sum = g.hasv;
    c->int_hsconst = c->d_acc[c->acc_parity].hs;
    c->int_vconst = nullptr;
    c->epilogue_fused = false;                           // set again below by a pass that finishes its own scores
    c->last_form = 0;
    if (needs_prep_dwords(c)) { const int rc = ensure_prep_dwords(c); if (rc) return rc; }
    if (!g.generic && shape_now(c) == 6) {
        if (!c->coef_ready) { const int rc = enqueue_bit_prep(c, true); if (rc) return rc; }
        has_hs_sum = g.nhs > 0 ? 1 : 0;
        if (c->mixed) {
            // Mixed layout: the value segment on the matrix cores (its bit tiles hold nothing else: the kernel leaves zeros in the
            // saturation rows of the partial sums), then the saturation byte planes with v_sad_u8 into those rows -- half the
            // vector work of the byte path's two sums -- and k_finish on both.  Off-level value patches take the kernel's int8 form
            // as always.
            c->fuse_request = false;
            launch_mfma(c, has_hs_sum);
            HIP_TRY(c, hipGetLastError());
            c->hs_bytes_pass = true;
            const int keep = c->force_shape;
            c->force_shape = c->APAD >= 16 ? 2 : 1;                       // four waves share an item and fold in LDS (tools/sweep_grid.sh)
            switch (g.nhs) {
                case 1: launch_tiles_apad<1, 0>(c); break;
                case 2: launch_tiles_apad<2, 0>(c); break;
                case 3: launch_tiles_apad<3, 0>(c); break;
                default: launch_tiles_apad<4, 0>(c); break;
            }
            c->force_shape = keep;
            c->hs_bytes_pass = false;
            c->nchunk = 1;
        } else
        launch_mfma(c, has_hs_sum);
        c->last_form = DV_FORM_MATRIX_CORES | (c->fp4_ok ? DV_FORM_FP4 : 0) | (c->epilogue_fused ? DV_FORM_FUSED_FINISH : 0);
        c->int_hsconst = c->mixed ? c->d_acc[c->acc_parity].hs : c->d_acc[c->acc_parity].bhs;
        c->int_vconst = c->d_acc[c->acc_parity].bv;
    } else if (g.generic) {
        has_hs_sum = 1;
        if (g.hasv) launch_generic_apad<1, 1>(c); else launch_generic_apad<1, 0>(c);
    } else {
        has_hs_sum = g.nhs > 0 ? 1 : 0;
        switch (g.nhs * 2 + g.hasv) {
            case 1: launch_tiles_apad<0, 1>(c); break;
            case 2: launch_tiles_apad<1, 0>(c); break;
            case 3: launch_tiles_apad<1, 1>(c); break;
            case 4: launch_tiles_apad<2, 0>(c); break;
            case 5: launch_tiles_apad<2, 1>(c); break;
            case 6: launch_tiles_apad<3, 0>(c); break;
            case 7: launch_tiles_apad<3, 1>(c); break;
            case 8: launch_tiles_apad<4, 0>(c); break;
            case 9: launch_tiles_apad<4, 1>(c); break;
            default: return fail(c, DV_ERR_STATE, "unsupported plane configuration nhs=%d hasv=%d", g.nhs, g.hasv);
        }
    }
    HIP_TRY(c, hipGetLastError());
    if (after_tiles) HIP_TRY(c, hipEventRecord(after_tiles, c->stream));
    c->int_has_hs = has_hs_sum;
    c->int_has_v = has_v_sum;
    if (c->epilogue_fused) return DV_OK;                 // the scoring kernel finished its scores itself: only k_fold is left
    if (!with_combine) return DV_OK;                     // the step ends in k_finish, which does the combining itself
    *n_partial = (int)((g.Fpad + 1023) / 1024);
    hipLaunchKernelGGL(k_combine, dim3((unsigned)*n_partial, (unsigned)c->A), dim3(256), 0, c->stream, c->d_part, c->int_hsconst,
                       c->int_vconst, c->d_fam, c->d_pmax, c->d_state, c->cfg, c->nchunk, c->APAD, has_hs_sum, has_v_sum, c->n_agents);
    HIP_TRY(c, hipGetLastError());
    return DV_OK;
}

// Times the workgroup shapes of k_sad_tiles (+ k_combine, whose input they size) on the resident library and patches
// and keeps the fastest for this heading class.  Runs once, on the first scoring pass after a library is set: four
// extra passes per shape and one host wait each.
static int tune_workgroup_shape(dv_ctx* c) {
    hipEvent_t ev[2];
    HIP_TRY(c, hipEventCreate(&ev[0]));
    HIP_TRY(c, hipEventCreate(&ev[1]));
    const int cls = apad_class(c->APAD);
    for (int i = 0; i < 6; ++i) c->tuned_us[cls][i] = 0.f;
    float best = 0.f;
    bool timed_any = false;
    int rc = DV_OK, pick = 1, np = 0;
    // the timing passes never finish their scores themselves: a fused epilogue appends candidates that only k_fold
    // clears, and the shapes are compared on scoring + k_combine
    const bool fuse_request = c->fuse_request;
    c->fuse_request = false;
    // The matrix-core pass (shape 6) is timed first; a byte-plane shape is then timed only if it could win at all: it
    // streams tile_bytes per pass, so it cannot take less than tile_bytes / 8 TB/s (500 000 views x 128x128: 2 ms against
    // 0.95 ms measured -- the ~16 byte-path launches of 4.7-5.7 ms each that used to open the first step are skipped).
    float mfma_ms = -1.f;
    const float byte_bound_ms = (float)((double)c->tile_bytes / 8.0e12 * 1e3);
    for (int k = 0; k < kMaxShape && rc == DV_OK; ++k) {
        const int sh = k == 0 ? kMaxShape : k;                        // 6, 1, 2, 3, 4, 5
        if (!shape_valid(c, cls, sh)) continue;
        if (sh != kMaxShape && mfma_ms >= 0.f && byte_bound_ms > mfma_ms && !c->tune_all_env) continue;
        c->force_shape = sh;
        rc = launch_int_scoring(c, nullptr, &np);                     // warm: code objects, caches
        if (rc == DV_OK && hipEventRecord(ev[0], c->stream) != hipSuccess) rc = DV_ERR_HIP;
        for (int k = 0; k < 3 && rc == DV_OK; ++k) rc = launch_int_scoring(c, nullptr, &np);
        if (rc == DV_OK && (hipEventRecord(ev[1], c->stream) != hipSuccess || hipEventSynchronize(ev[1]) != hipSuccess))
            rc = DV_ERR_HIP;
        float ms = 0.f;
        if (rc == DV_OK && hipEventElapsedTime(&ms, ev[0], ev[1]) != hipSuccess) rc = DV_ERR_HIP;
        if (rc == DV_OK) {
            c->tuned_us[cls][sh - 1] = ms * 1e3f / 3.f;
            if (!timed_any || ms < best) { best = ms; pick = sh; }
            timed_any = true;
            if (sh == kMaxShape) mfma_ms = ms / 3.f;
        }
    }
    c->force_shape = 0;
    c->fuse_request = fuse_request;
    c->epilogue_fused = false;
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    if (rc == DV_ERR_HIP) return fail(c, DV_ERR_HIP, "timing the workgroup shapes failed: %s", hipGetErrorString(hipGetLastError()));
    if (rc) return rc;
    c->tuned_shape[cls] = pick;
    if (getenv("DEJAVU_VERBOSE")) {
        fprintf(stderr, "[dejavu] %d resident headings, us per scoring pass by workgroup shape:", c->APAD);
        for (int sh = 1; sh <= kMaxShape; ++sh)
            if (shape_valid(c, cls, sh) && c->tuned_us[cls][sh - 1] > 0.f) fprintf(stderr, " %d: %.1f", sh, c->tuned_us[cls][sh - 1]);
        fprintf(stderr, " -> shape %d\n", pick);
    }
    return DV_OK;
}

// Scoring: integer-sum kernel + combine (or the exact fp64 kernel), then amax[a] without atomics.
static int launch_scoring(dv_ctx* c, bool with_combine = true) {
    Range range("dv:score");
    const LibCfg& g = c->cfg;
    int rc = DV_OK;
    if (c->metric == 0 && !c->exact && !g.generic && c->shape_env == 0 && c->tuned_shape[apad_class(c->APAD)] == 0) {
        rc = tune_workgroup_shape(c);
        if (rc) return rc;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    const bool prof = c->profile > 0 && (c->profile_count++ % c->profile) == 0;
    if (prof) {
        if (c->pev_used + 2 > c->pev.size()) {
            for (int i = 0; i < 2; ++i) {
                hipEvent_t e;
                HIP_TRY(c, hipEventCreate(&e));
                c->pev.push_back(e);
            }
        }
        e0 = c->pev[c->pev_used];
        e1 = c->pev[c->pev_used + 1];
        c->pev_used += 2;
        HIP_TRY(c, hipEventRecord(e0, c->stream));
    }
    int n_partial = 0;
    if (c->metric == 2) {
        // one pass of 32 headings at a time over the whole library; a workgroup of 8 waves per CU (its operand rows take up to 128 KB
        // of LDS), two view groups per wave where the library fills every CU with such ranges, one below that; the ranges cut evenly
        static bool attr_set = false;
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)k_ssd_u8_mfma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            (void)hipFuncSetAttribute((const void*)k_ssd_u8_mfma<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
            attr_set = true;
        }
        const long long G32 = g.Fpad / 32;
        const int tl = G32 >= 256 * 16 ? 2 : 1;
        long long n_items = (G32 + 8 * tl - 1) / (8 * tl);
        if (n_items < 256) n_items = G32 < 256 ? G32 : 256;
        else n_items = (n_items + 255) / 256 * 256;                    // whole rounds of the 256 workgroups, every range the same size
        const unsigned grid = (unsigned)(n_items < 256 ? n_items : 256);
        const size_t lds = (size_t)c->u8_KC * 1024;
        for (int a_off = 0; a_off < c->APAD; a_off += 32) {
            const uint4* rows = c->d_u8prep + (size_t)(a_off / 32) * c->u8_K * 64;
            if (tl == 2)
                hipLaunchKernelGGL((k_ssd_u8_mfma<2>), dim3(grid), dim3(512), lds, c->stream, c->d_u8tiles, rows, c->d_u8part, c->cfg, c->u8_K, c->u8_KC,
                                   c->u8_nchunk, c->APAD, a_off, n_items);
            else
                hipLaunchKernelGGL((k_ssd_u8_mfma<1>), dim3(grid), dim3(512), lds, c->stream, c->d_u8tiles, rows, c->d_u8part, c->cfg, c->u8_K, c->u8_KC,
                                   c->u8_nchunk, c->APAD, a_off, n_items);
        }
        HIP_TRY(c, hipGetLastError());
        if (prof) HIP_TRY(c, hipEventRecord(e1, c->stream));
        n_partial = (int)(g.Fpad / 256) + ((g.Fpad % 256) ? 1 : 0);
        hipLaunchKernelGGL(k_combine_u8, dim3((unsigned)n_partial, (unsigned)c->A), dim3(256), 0, c->stream, c->d_u8part, c->d_vnorm, c->d_pnorm,
                           c->d_fam, c->d_pmax, c->d_state, c->cfg, c->u8_nchunk, c->APAD, c->n_agents);
        HIP_TRY(c, hipGetLastError());
        c->n_partial = n_partial;
        return DV_OK;
    }
    if (c->metric == 1) {
        if (c->exact) {
            hipLaunchKernelGGL(k_exact_all_f32, dim3((unsigned)(g.Fpad / 64), (unsigned)((c->A + 3) / 4)), dim3(64, 4), 0, c->stream,
                               c->d_ftiles, c->d_fraw, c->d_fam, c->d_pmax, c->d_state, c->cfg, c->A, c->n_agents);
            HIP_TRY(c, hipGetLastError());
            n_partial = (int)(g.Fpad / 64);
        } else if (c->f32x_request && c->ssd_mfma_env) {
            // The cross-term form on the matrix cores (k_ssd_f32_mfma): 32 headings per pass over the library (16 when no more are
            // resident), items = (pixel chunk, view group) on single-wave workgroups as many as are resident at once.
            auto launch = [&](auto kern, int hb, int a_off) {
                dim3 block;
                const dim3 grid = scoring_grid(c, resident_waves_per_cu((const void*)kern), block);
                hipLaunchKernelGGL(kern, grid, block, 0, c->stream, c->d_ftiles, c->d_fprep4, c->d_fpart, c->cfg, c->nchunk, c->APAD, a_off);
                (void)hb;
            };
            // up to 16 resident headings: the 16-wide fp32 instruction (bound by its stream); passes of 32: the two-term bf16 form
            // (DEJAVU_SSD_MFMA=2: the 32-wide fp32 instruction instead, 67 % matrix-pipe-bound at 1.79 GHz)
            c->f32x_kappa = kF32xKappa;
            if (c->APAD <= 16 && c->ssd_mfma_env != 3) launch(k_ssd_f32_mfma<16, 8>, 16, 0);
            else if (c->ssd_mfma_env == 2) for (int a_off = 0; a_off < c->APAD; a_off += 32) launch(k_ssd_f32_mfma<32, 16>, 32, a_off);
            else {
                c->f32x_kappa = kF32xKappaBf16;
                for (int a_off = 0; a_off < c->APAD; a_off += 32) {                    // (DEJAVU_SSD_MFMA=3: also for <= 16 headings, A/B)
                    dim3 block;
                    const dim3 grid = scoring_grid(c, resident_waves_per_cu((const void*)k_ssd_f32_bf16x2<4>), block);
                    hipLaunchKernelGGL(k_ssd_f32_bf16x2<4>, grid, block, 0, c->stream, c->d_ftiles, c->d_fprepb, c->d_fpart, c->cfg, c->nchunk, c->APAD, a_off);
                }
            }
            HIP_TRY(c, hipGetLastError());
            if (prof) HIP_TRY(c, hipEventRecord(e1, c->stream));
            n_partial = (int)(g.Fpad / 256) + ((g.Fpad % 256) ? 1 : 0);
            hipLaunchKernelGGL(k_combine_f32x, dim3((unsigned)n_partial, (unsigned)c->A), dim3(256), 0, c->stream, c->d_fpart, c->d_fvnorm,
                               c->d_fpnorm, c->d_fam, c->d_flower, c->d_state, c->cfg, c->nchunk, c->APAD, c->n_agents, c->f32x_kappa);
            HIP_TRY(c, hipGetLastError());
            c->f32x_used = true;
        } else {
            // 16 headings per pass (8 when no more are resident).  DEJAVU_SHAPE: 1 single-wave workgroups, 2 four waves
            // + LDS fold, 3 / 4 the same with the next block's tiles prefetched.  Measured on 50 000 views x 64x64 x 16
            // headings (819 MB): 191 / 308 / 181 / 207 us -> default 3.
            { const int rc2 = ensure_direct_prep_f32(c); if (rc2) return rc2; }
            const int var = (c->shape_env >= 1 && c->shape_env <= 4) ? c->shape_env : 3;
            auto launch = [&](auto kern, int nw, int apad, int a_off) {
                static_assert(true, "");
                dim3 block;
                const size_t lds = nw > 1 ? (size_t)nw * apad * 64 * sizeof(double) : 0;
                const dim3 grid = scoring_grid(c, resident_waves_per_cu((const void*)kern), block, nw, lds);
                hipLaunchKernelGGL(kern, grid, block, lds, c->stream, c->d_ftiles, c->d_fprep, c->d_fpart, c->cfg, c->nchunk, c->APAD, a_off);
            };
            if (c->APAD == 8) {
                if (var == 1) launch(k_ssd_tiles<8, 1, false>, 1, 8, 0);
                else if (var == 3) launch(k_ssd_tiles<8, 1, true>, 1, 8, 0);
                else if (var == 4) launch(k_ssd_tiles<8, 4, true>, 4, 8, 0);
                else launch(k_ssd_tiles<8, 4, false>, 4, 8, 0);
            } else {
                for (int a_off = 0; a_off < c->APAD; a_off += 16) {
                    if (var == 1) launch(k_ssd_tiles<16, 1, false>, 1, 16, a_off);
                    else if (var == 3) launch(k_ssd_tiles<16, 1, true>, 1, 16, a_off);
                    else if (var == 4) launch(k_ssd_tiles<16, 4, true>, 4, 16, a_off);
                    else launch(k_ssd_tiles<16, 4, false>, 4, 16, a_off);
                }
            }
            HIP_TRY(c, hipGetLastError());
            if (prof) HIP_TRY(c, hipEventRecord(e1, c->stream));
            n_partial = (int)(g.Fpad / 256) + ((g.Fpad % 256) ? 1 : 0);
            hipLaunchKernelGGL(k_combine_f32, dim3((unsigned)n_partial, (unsigned)c->A), dim3(256), 0, c->stream, c->d_fpart,
                               c->d_fam, c->d_pmax, c->d_state, c->cfg, c->nchunk, c->APAD, c->n_agents);
            HIP_TRY(c, hipGetLastError());
        }
        if (c->exact && prof) HIP_TRY(c, hipEventRecord(e1, c->stream));
        c->n_partial = n_partial;
        return DV_OK;
    }
    if (c->exact) {
        hipLaunchKernelGGL(k_exact_all, dim3((unsigned)(g.Fpad / 64), (unsigned)((c->A + 3) / 4)), dim3(64, 4), 0, c->stream,
                           c->d_tiles, c->d_raw_patches, c->d_fam, c->d_pmax, c->d_state, c->cfg, c->A, c->n_agents);
        HIP_TRY(c, hipGetLastError());
        n_partial = (int)(g.Fpad / 64);
    } else {
        rc = launch_int_scoring(c, prof ? e1 : nullptr, &n_partial, with_combine);
        if (rc) return rc;
    }
    if (c->exact && prof) HIP_TRY(c, hipEventRecord(e1, c->stream));
    c->n_partial = n_partial;
    return DV_OK;
}

// k_sense's / k_patch_prep's "ran off the landscape" word of the resident patches (nullptr: they were not sensed).
static const unsigned long long* sense_err_ptr(const dv_ctx* c) {
    if (!c->patches_sensed) return nullptr;
    return c->metric == 2 ? c->d_sense_err : &c->d_acc[c->acc_parity].err;
}

// The arrival ticket of k_finish / k_tail without fences saves ~1.5 us of a ~100 us single-agent step; everywhere else
// (exact scores, ssd_f32, batched passes) the release / acquire pair of the memory model is kept.
static int step_fenced(const dv_ctx* c) {
    if (c->fenced_env >= 0) return c->fenced_env;
    if (c->exact || c->metric != 0 || c->n_agents > 1) return 1;
    // single-agent integer path: unfenced only where the ~1.5 us pair is a visible share of the step -- scoring passes
    // estimated under 200 us (bytes streamed at 4 TB/s; the byte path and the mixed layout at 500 000 views x 128x128 take
    // 3-5 ms and are fenced)
    const double bytes = (c->bits_ok && !c->mixed) ? (double)c->btile_bytes : (double)c->tile_bytes;
    return bytes / 4.0e12 >= 200e-6 ? 1 : 0;
}

// One step on the resident patches: scoring (2 launches) + k_tail.  The result record lands in mapped host memory.
// k_fold behind summaries left by k_finish or by a fused scoring epilogue.  Long lists (more than 512 summaries per
// agent) are first cut down to kFoldSlices by k_fold_reduce on as many workgroups: one workgroup walking 977 summaries x
// 32 headings took 23.5 us at 500 000 views.
static void launch_fold(dv_ctx* c, int nb, StepResultDev* outp, double* recp, int force, int seq, const unsigned long long* serr) {
    unsigned long long* sums = c->d_bsum;
    if (nb > 512) {
        const int per = (nb + kFoldSlices - 1) / kFoldSlices;
        const int slices = (nb + per - 1) / per;
        hipLaunchKernelGGL(k_fold_reduce, dim3((unsigned)slices, (unsigned)c->n_agents), dim3(256), 0, c->stream, c->d_bsum, c->d_bsum2,
                           c->d_ctmp, c->d_state, c->A_agent, c->delta, nb, per);
        sums = c->d_bsum2;
        nb = slices;
    }
    // threads: enough that every thread walks at most 16 summaries of its heading in one round trip (fold_and_decide), no more
    // -- a 1024-thread workgroup spends its time in its own barriers when 32 summaries are left
    int threads = 1024;
    if ((long long)nb * c->A_agent <= 256ll * 16) threads = 256;
    else if ((long long)nb * c->A_agent <= 512ll * 16) threads = 512;
    const dim3 grid(1, (unsigned)c->n_agents);
    if (threads == 256)
        hipLaunchKernelGGL(k_fold<256>, grid, dim3(256), 0, c->stream, sums, c->d_ctmp, c->d_cand, c->d_state, outp, recp, c->cfg, c->A_agent,
                           c->delta, force, seq, serr, nb);
    else if (threads == 512)
        hipLaunchKernelGGL(k_fold<512>, grid, dim3(512), 0, c->stream, sums, c->d_ctmp, c->d_cand, c->d_state, outp, recp, c->cfg, c->A_agent,
                           c->delta, force, seq, serr, nb);
    else
        hipLaunchKernelGGL(k_fold<1024>, grid, dim3(1024), 0, c->stream, sums, c->d_ctmp, c->d_cand, c->d_state, outp, recp, c->cfg, c->A_agent,
                           c->delta, force, seq, serr, nb);
}

template <int NT>
static void launch_finish(dv_ctx* c, int want_scene, int force) {
    const LibCfg& g = c->cfg;
    // Large libraries (more than 256 blocks of 256 views): the blocks only leave their summaries and k_fold, a 1024-thread
    // kernel behind them, folds and decides -- the serial walk of thousands of summaries by one 256-thread last block cost
    // 60 us at 500 000 views.  DEJAVU_FINISH_VB: view sets of 256 per block (fewer, longer blocks; default 1).
    long long vb = c->finish_vb_env > 0 ? c->finish_vb_env : 1;
    const long long per_block = 256 * vb;
    const unsigned nb = (unsigned)((g.F + per_block - 1) / per_block);
    const int separate = nb > 256 ? 1 : 0;
    const unsigned long long* serr = sense_err_ptr(c);
    StepResultDev* outp = c->d_result + c->result_slot;
    double* recp = c->d_record + (size_t)c->result_slot * (3 + 4 * kMaxHeadings);
    ++c->seq;
    hipLaunchKernelGGL(k_finish<NT>, dim3(nb, (unsigned)c->n_agents), dim3(256), 0, c->stream,
                       c->d_part, c->int_hsconst, c->int_vconst, c->nchunk, c->APAD, c->int_has_hs, c->int_has_v, c->d_state, c->d_bsum, c->d_ctmp,
                       c->d_cand, c->d_scene, outp, recp, c->cfg, c->A_agent, c->delta, want_scene, force,
                       c->seq, serr, step_fenced(c), (int)vb, separate);
    if (separate) launch_fold(c, (int)nb, outp, recp, force, c->seq, serr);
}

static int enqueue_step(dv_ctx* c, uint32_t flags, bool want_scene) {
    const LibCfg& g = c->cfg;
    c->agent_pending = 0;                                               // (dv_agent_step_begin sets it behind its own enqueue)
    const int force = (flags & DV_STEP_FORCE_RESOLVE) ? 1 : 0;
    const int scene_on = (want_scene && c->n_agents == 1) ? 1 : 0;
    // integer path: the scoring kernel's partial sums go straight to k_finish (combine + reductions + decision in one
    // launch); the exact mode and ssd_f32 produce fam[] first and end in k_tail
    // Where it pays (measured, DEJAVU_FINISH=0/1/2): up to 16 headings per agent and libraries from ~32 k views.
    // 32 headings per agent: equal at 200 000 views, 5 us slower at 20 000; 64 (64 scores per thread in registers,
    // one wave per SIMD): 1103 vs 1029 us at 200 000 views; 8-16 headings on 20 000 views: 1-2 us slower.
    const bool fused = c->metric == 0 && !c->exact && c->A_agent <= 32 &&
                       (c->finish_fused == 2 || (c->finish_fused == 1 && g.F >= 32768 && (c->A_agent <= 16 || g.F >= 131072)));
    // With one chunk the matrix-core kernel can finish its scores in its own epilogue (k_sad_mfma_ring<.., FUSE>): the
    // partial sums never reach HBM and the step ends in k_fold alone.  Not with scene_fam: its minimum over headings
    // runs across the lanes there.
    c->fuse_request = c->metric == 0 && !c->exact && !want_scene;
    c->epilogue_fused = false;
    // ssd_f32 without per-view output: the cross-term form on the matrix cores selects, the exact resolver scores (k_ssd_f32_mfma)
    c->f32x_request = c->metric == 1 && !c->exact && !want_scene && c->n_agents == 1;
    c->f32x_used = false;
    int rc = launch_scoring(c, !fused);
    c->fuse_request = false;
    c->f32x_request = false;
    if (rc) return rc;
    Range range("dv:finish");
    if (c->f32x_used) {
        // candidates per heading -> exact re-scoring -> minima and decision from the exact values; the record is k_decide's
        hipLaunchKernelGGL(k_cand_f32x, dim3((unsigned)((g.F + 255) / 256)), dim3(256), 0, c->stream, c->d_fam, c->d_flower,
                           c->d_fvnorm, c->d_fpnorm, c->d_state, c->d_cand, c->cfg, c->A_agent, c->f32x_kappa);
        hipLaunchKernelGGL(k_resolve_f32, dim3(256), dim3(64), 0, c->stream, c->d_ftiles, c->d_fraw, c->d_state, c->d_cand, c->d_cand_exact, c->cfg);
        hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, c->stream, c->d_state, c->d_cand, c->d_cand_exact, c->d_result + c->result_slot,
                           c->d_record + (size_t)c->result_slot * (3 + 4 * kMaxHeadings), c->cfg, c->A_agent, c->delta, sense_err_ptr(c), 0,
                           ++c->seq);
    } else if (c->epilogue_fused && c->defer_fold) {
        c->deferred_force = force;                           // run_batch launches this pass's fold behind the group's scoring kernels
        c->deferred_seq = ++c->seq;
    } else if (c->epilogue_fused) {
        launch_fold(c, c->fused_nb, c->d_result + c->result_slot, c->d_record + (size_t)c->result_slot * (3 + 4 * kMaxHeadings), force, ++c->seq,
                    sense_err_ptr(c));
    } else if (fused) {
        if (c->A_agent <= 16) launch_finish<1>(c, scene_on, force);
        else launch_finish<2>(c, scene_on, force);
    } else {
        // (one view per thread: with 64 blocks walking the views block-stride -- fewer arrival tickets on the one word -- k_tail took
        // 18.5 us instead of 14.9 at 50 000 views x 16 headings: its time is the scores' read, not the tickets)
        hipLaunchKernelGGL(k_tail, dim3((unsigned)((g.F + 255) / 256), (unsigned)c->n_agents), dim3(256), 0, c->stream, c->d_fam,
                           c->d_pmax, c->n_partial, c->d_state, c->d_cand, c->d_scene, c->d_result + c->result_slot,
                           c->d_record + (size_t)c->result_slot * (3 + 4 * kMaxHeadings), c->cfg,
                           c->A_agent, c->delta, scene_on, (c->exact || c->metric == 2) ? 1 : 0, force, ++c->seq,
                           sense_err_ptr(c), c->metric == 1 ? 3e-6 : 0.0, step_fenced(c));
    }
    HIP_TRY(c, hipGetLastError());
    if (want_scene)
        HIP_TRY(c, hipMemcpyAsync(c->h_scene, c->d_scene, (size_t)g.F * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    c->step_pending = true;
    c->last_want_scene = want_scene;
    return DV_OK;
}

// Rare path: re-score one agent's listed candidates with the exact kernel and decide on those values.
static int enqueue_resolve(dv_ctx* c, int agent = 0) {
    if (c->metric == 2) return DV_OK;                      // ssd_u8: every score is the exact integer already, nothing to re-score
    const size_t co = (size_t)agent * kCandCap;
    if (c->metric == 1)
        hipLaunchKernelGGL(k_resolve_f32, dim3(256), dim3(64), 0, c->stream, c->d_ftiles,
                           c->d_fraw + (size_t)agent * c->A_agent * c->cfg.P, c->d_state + agent, c->d_cand + co,
                           c->d_cand_exact + co, c->cfg);
    else
    hipLaunchKernelGGL(k_resolve, dim3(256), dim3(64), 0, c->stream, c->d_tiles,
                       c->d_raw_patches + (size_t)agent * c->A_agent * c->cfg.P * 3, c->d_state + agent, c->d_cand + co,
                       c->d_cand_exact + co, c->cfg);
    HIP_TRY(c, hipGetLastError());
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(64), 0, c->stream, c->d_state + agent, c->d_cand + co, c->d_cand_exact + co,
                       c->d_result + agent, c->d_record + (size_t)agent * (3 + 4 * kMaxHeadings), c->cfg, c->A_agent, c->delta,
                       sense_err_ptr(c), agent, -1);
    HIP_TRY(c, hipGetLastError());
    return DV_OK;
}

extern "C" int dv_step_enqueue(dv_ctx* c, uint32_t flags) {
    if (!c) return DV_ERR_INVALID;
    if (!c->have_lib || c->A < 1) return fail(c, DV_ERR_STATE, "no library or no resident patches");
    HIP_TRY(c, hipSetDevice(c->device));
    return enqueue_step(c, flags, (flags & DV_STEP_WANT_SCENE) != 0);
}

// Waits for the enqueued pass; runs the exact resolver for agents whose near-ties need it; redoes the pass with exact
// scores if a candidate list overflowed.  Results are left in c->h_result[0 .. n_agents).
// Host wait for k_tail: the last word of every agent's record (n_headings | seq << 32) is stored after a
// system-scope release, so once all agents show the current sequence number their records are complete.
// Polls result records [slot, slot + n) until each shows sequence number `seq` and a matching check word.
static bool spin_for_records(dv_ctx* c, int slot, int n, int A, int seq) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int ag = 0; ag < n; ++ag) {
        volatile const int* word = &c->h_result[slot + ag].reserved;
        volatile const unsigned long long* w = reinterpret_cast<volatile const unsigned long long*>(&c->h_result[slot + ag]);
        unsigned spins = 0;
        for (;;) {
            if (*word == seq) {
                // the record is stored without a fence: take it only when its check word agrees with its words
                std::atomic_thread_fence(std::memory_order_acquire);
                unsigned long long snap[7 + 4 * kMaxHeadings];
                for (int i = 0; i < 7; ++i) snap[i] = w[i];
                for (int k = 0; k < 4; ++k)
                    for (int a = 0; a < A; ++a) snap[7 + k * kMaxHeadings + a] = w[7 + k * kMaxHeadings + a];
                const unsigned long long check = *reinterpret_cast<volatile const unsigned long long*>(&c->h_result[slot + ag].check);
                if ((int)(snap[6] >> 32) == seq && record_check(snap, A) == check) break;
            }
            if ((++spins & 1023u) == 0 &&
                std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) return false;   // fall back
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    return true;
}

static bool spin_for_results(dv_ctx* c) { return spin_for_records(c, 0, c->n_agents, c->A_agent, c->seq); }

static int finish_pass(dv_ctx* c) {
    Range range("dv:wait");
    if (!c->step_pending) return fail(c, DV_ERR_STATE, "no step enqueued");
    if (!(c->spin_wait && !c->last_want_scene && spin_for_results(c)))
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    bool any = false;
    for (int ag = 0; ag < c->n_agents; ++ag) {
        if (c->h_result[ag].flags & kResNeedsResolve) {
            // two or more (heading, view) pairs within delta of the maximum: the exact kernel decides
            int rc = enqueue_resolve(c, ag);
            if (rc) return rc;
            any = true;
        }
    }
    if (any) HIP_TRY(c, hipStreamSynchronize(c->stream));
    bool overflow = false;
    for (int ag = 0; ag < c->n_agents; ++ag) overflow |= (c->h_result[ag].flags & DV_RES_OVERFLOW) != 0;
    if (overflow) {
        // More near-ties than a candidate list holds: redo this pass with exact scores everywhere.
        const int was_exact = c->exact;
        std::vector<long long> n_first(c->n_agents);
        std::vector<unsigned> had(c->n_agents);
        for (int ag = 0; ag < c->n_agents; ++ag) { n_first[ag] = c->h_result[ag].n_candidates; had[ag] = c->h_result[ag].flags; }
        c->exact = 1;
        int rc = enqueue_step(c, 0, c->last_want_scene);
        c->exact = was_exact;
        if (rc) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int ag = 0; ag < c->n_agents; ++ag)
            if (had[ag] & DV_RES_OVERFLOW) { c->h_result[ag].flags |= DV_RES_OVERFLOW; c->h_result[ag].n_candidates = n_first[ag]; }
    }
    return DV_OK;
}

static void copy_result(const dv_ctx* c, int agent, dv_step_result* result) {
    memcpy(result, &c->h_result[agent], sizeof *result);
    for (int a = c->A_agent; a < kMaxHeadings; ++a) {
        result->angle_fam[a] = 0.0; result->angle_view[a] = -1;
        result->exact_fam[a] = 0.0; result->exact_view[a] = -1;
    }
}

static int wait_step(dv_ctx* c, dv_step_result* result, double* scene_fam) {
    int rc = finish_pass(c);
    if (rc) return rc;
    if (c->h_result[0].flags & kResSenseError)
        return fail(c, DV_ERR_INDEX, "sensor footprint reaches past the end of the landscape (index out of bounds)");
    if (result) copy_result(c, 0, result);
    if (scene_fam) {
        if (!c->last_want_scene) return fail(c, DV_ERR_STATE, "scene familiarity was not requested for this step");
        memcpy(scene_fam, c->h_scene, (size_t)c->cfg.F * sizeof(double));
    }
    return DV_OK;
}

extern "C" int dv_step_wait(dv_ctx* c, dv_step_result* result, double* scene_fam) {
    if (!c) return DV_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    return wait_step(c, result, scene_fam);
}

extern "C" int dv_step(dv_ctx* c, const uint8_t* patches, int A, uint32_t flags, dv_step_result* result,
                       double* scene_fam) {
    int rc = dv_upload_patches(c, patches, A);
    if (rc) return rc;
    if (!result) return fail(c, DV_ERR_INVALID, "result is NULL");
    rc = enqueue_step(c, flags, scene_fam != nullptr);
    if (rc) return rc;
    return wait_step(c, result, scene_fam);
}

// Batched agents (ensemble runs): N agents x A headings each against the same library.  Agents are grouped into
// passes of at most DV_MAX_HEADINGS headings; one pass = one scoring launch set, one k_tail with an agent per
// blockIdx.y.  Each agent's decision obeys the same rules as dv_step.
extern "C" int dv_step_batch(dv_ctx* c, const uint8_t* patches, int n_agents, int A, uint32_t flags, dv_step_result* results) {
    int rc = check_step_args(c, A);
    if (rc) return rc;
    if (!patches || !results || n_agents < 1) return fail(c, DV_ERR_INVALID, "dv_step_batch: bad arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t agent_bytes = (size_t)A * c->cfg.P * 3;
    return run_batch(c, n_agents, A, flags, results, [&](int first, int n) {
        if (hipMemcpyAsync(c->d_raw_patches, patches + (size_t)first * agent_bytes, (size_t)n * agent_bytes, hipMemcpyHostToDevice,
                           c->stream) != hipSuccess)
            return fail(c, DV_ERR_HIP, "dv_step_batch: patch upload failed: %s", hipGetErrorString(hipGetLastError()));
        const int rc2 = prep_patches(c, n * A);
        if (rc2) return rc2;
        c->n_agents = n;
        c->A_agent = A;
        return (int)DV_OK;
    }, true);
}

// Steps of any number of headings: passes of at most kMaxHeadings, merged by the rule of NavBySceneFamiliarity.py:313-315.
// `stage(first, n)` makes headings [first, first + n) the resident patches.  Without scene_fam the passes are enqueued back to
// back (each with its own result record, `result_slot`) and collected afterwards; a pass whose near-ties need the exact resolver
// or whose candidate list overflowed is run again on its own through the synchronous path, as run_batch does.
template <class Stage>
static int run_wide(dv_ctx* c, int A, uint32_t flags, dv_wide_result* out, double* angle_fam, int64_t* angle_view, double* scene_fam,
                    Stage stage) {
    struct Pass { int first, n, seq; dv_step_result r; };
    const int npass = (A + kMaxHeadings - 1) / kMaxHeadings;
    std::vector<Pass> ps((size_t)npass);
    for (int p = 0; p < npass; ++p) {
        ps[p].first = p * kMaxHeadings;
        ps[p].n = A - ps[p].first < kMaxHeadings ? A - ps[p].first : kMaxHeadings;
    }
    int rc = DV_OK;
    auto index_error = [&]() { return fail(c, DV_ERR_INDEX, "sensor footprint reaches past the end of the landscape (index out of bounds)"); };
    auto run_sync = [&](Pass& p, uint32_t fl, double* scene) -> int {
        int r2 = stage(p.first, p.n);
        if (r2) return r2;
        c->result_slot = 0;
        r2 = enqueue_step(c, fl, scene != nullptr);
        if (r2) return r2;
        r2 = finish_pass(c);
        if (r2) return r2;
        if (c->h_result[0].flags & kResSenseError) return index_error();
        copy_result(c, 0, &p.r);
        if (scene) memcpy(scene, c->h_scene, (size_t)c->cfg.F * sizeof(double));
        return DV_OK;
    };
    if (scene_fam) {
        // the per-view minimum over ALL headings (:301-303): pass by pass, merged on the host (the reference only plots it)
        std::vector<double> tmp;
        for (int p = 0; p < npass; ++p) {
            if (p == 1) tmp.resize((size_t)c->cfg.F);
            rc = run_sync(ps[p], flags, p == 0 ? scene_fam : tmp.data());
            if (rc) return rc;
            if (p > 0) for (int64_t f = 0; f < c->cfg.F; ++f) scene_fam[f] = tmp[(size_t)f] < scene_fam[f] ? tmp[(size_t)f] : scene_fam[f];
        }
    } else {
        for (int p = 0; p < npass; ++p) {
            rc = stage(ps[p].first, ps[p].n);
            if (rc) break;
            c->result_slot = p;
            rc = enqueue_step(c, flags, false);
            ps[p].seq = c->seq;
            if (rc) break;
        }
        c->result_slot = 0;
        if (rc) { (void)hipStreamSynchronize(c->stream); return rc; }
        bool polled = c->spin_wait != 0;
        for (int p = 0; p < npass; ++p)
            if (polled && !spin_for_records(c, p, 1, ps[p].n, ps[p].seq)) polled = false;
        if (!polled) HIP_TRY(c, hipStreamSynchronize(c->stream));
        std::vector<int> again;
        for (int p = 0; p < npass; ++p) {
            const unsigned fl = c->h_result[p].flags;
            if (fl & kResSenseError) return index_error();
            if (fl & (kResNeedsResolve | DV_RES_OVERFLOW)) { again.push_back(p); continue; }
            memcpy(&ps[p].r, &c->h_result[p], sizeof(dv_step_result));
        }
        for (int p : again) {
            rc = run_sync(ps[p], flags, nullptr);
            if (rc) return rc;
        }
    }
    // merge: the passes whose maximum lies within the candidate window of the overall one contend; with more than one of them
    // every contender's candidates are re-scored exactly (forced resolver) and the exact maxima compared, first pass on ties --
    // passes are in heading order, so this is the first heading attaining the maximum
    double M = -std::numeric_limits<double>::infinity();
    for (const Pass& p : ps) M = p.r.approx_max > M ? p.r.approx_max : M;
    const double window = c->delta + (c->metric == 1 ? 3e-6 * std::fabs(M) : 0.0);
    std::vector<int> cont;
    for (int p = 0; p < npass; ++p) if (ps[p].r.approx_max >= M - window) cont.push_back(p);
    if (cont.size() > 1)
        for (int p : cont)
            if (!(ps[p].r.flags & (DV_RES_RESOLVED | DV_RES_EXACT_ALL))) {
                rc = run_sync(ps[p], flags | DV_STEP_FORCE_RESOLVE, nullptr);
                if (rc) return rc;
            }
    int win = cont[0];
    for (int p : cont) if (ps[p].r.best_fam > ps[win].r.best_fam) win = p;
    const dv_step_result& w = ps[win].r;
    out->best_heading = ps[win].first + w.best_heading;
    out->flags = w.flags & ~kResNeedsResolve;
    out->best_view = w.best_view;
    out->best_fam = w.best_fam;
    out->n_headings = A;
    out->n_passes = npass;
    out->n_contending = (int)cont.size();
    out->reserved = 0;
    for (const Pass& p : ps)
        for (int a = 0; a < p.n; ++a) {
            angle_fam[p.first + a] = p.r.angle_fam[a];
            if (angle_view) angle_view[p.first + a] = p.r.angle_view[a];
        }
    return DV_OK;
}

static int check_wide_args(dv_ctx* c, int A, const dv_wide_result* result, const double* angle_fam) {
    if (!c) return DV_ERR_INVALID;
    if (!c->have_lib) return fail(c, DV_ERR_STATE, "no library set (call dv_set_library first)");
    if (c->metric != 0) return fail(c, DV_ERR_STATE, "the wide steps score sads_hsv libraries");
    if (A < 1 || A > DV_MAX_WIDE_HEADINGS) return fail(c, DV_ERR_INVALID, "n_headings %d outside [1, %d]", A, DV_MAX_WIDE_HEADINGS);
    if (!result || !angle_fam) return fail(c, DV_ERR_INVALID, "result or angle_fam is NULL");
    return DV_OK;
}

extern "C" int dv_step_wide(dv_ctx* c, const uint8_t* patches, int A, uint32_t flags, dv_wide_result* result, double* angle_fam,
                            int64_t* angle_view, double* scene_fam) {
    int rc = check_wide_args(c, A, result, angle_fam);
    if (rc) return rc;
    if (!patches) return fail(c, DV_ERR_INVALID, "patches is NULL");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t heading_bytes = (size_t)c->cfg.P * 3;
    return run_wide(c, A, flags, result, angle_fam, angle_view, scene_fam, [&](int first, int n) {
        if (hipMemcpyAsync(c->d_raw_patches, patches + (size_t)first * heading_bytes, (size_t)n * heading_bytes, hipMemcpyHostToDevice,
                           c->stream) != hipSuccess)
            return fail(c, DV_ERR_HIP, "dv_step_wide: patch upload failed: %s", hipGetErrorString(hipGetLastError()));
        return prep_patches(c, n);
    });
}

extern "C" int dv_sense_step_wide(dv_ctx* c, double x, double y, const double* angles, int A, uint32_t flags, dv_wide_result* result,
                                  double* angle_fam, int64_t* angle_view, double* scene_fam) {
    int rc = check_wide_args(c, A, result, angle_fam);
    if (rc) return rc;
    if (!angles) return fail(c, DV_ERR_INVALID, "angles is NULL");
    if (!c->have_sensor) return fail(c, DV_ERR_STATE, "sensor not configured");
    if (c->sensor.sw != c->w || c->sensor.sh != c->h)
        return fail(c, DV_ERR_STATE, "sensor is %dx%d but the library holds %dx%d views", c->sensor.sw, c->sensor.sh, c->w, c->h);
    HIP_TRY(c, hipSetDevice(c->device));
    return run_wide(c, A, flags, result, angle_fam, angle_view, scene_fam, [&](int first, int n) {
        PoseSet poses;
        for (int a = 0; a < n; ++a) poses.p[a] = make_pose(x, y, angles[first + a]);
        for (int a = n; a < kMaxHeadings; ++a) poses.p[a] = Pose{0., 0., 1., 0.};
        return sense_prep_launch(c, poses, 1, n);
    });
}

extern "C" int dv_resolve(dv_ctx* c, dv_step_result* result) {
    if (!c || !result) return DV_ERR_INVALID;
    if (!c->have_lib || !c->step_pending) return fail(c, DV_ERR_STATE, "no step to resolve");
    HIP_TRY(c, hipSetDevice(c->device));
    if (c->metric != 2 && !(c->h_result->flags & (DV_RES_EXACT_ALL | DV_RES_RESOLVED))) {
        int rc = enqueue_resolve(c);
        if (rc) return rc;
    }
    return wait_step(c, result, nullptr);
}

extern "C" int dv_step_record(dv_ctx* c, void** device_ptr, int* n_doubles) {
    if (!c || !device_ptr || !n_doubles) return DV_ERR_INVALID;
    if (!c->have_lib || c->A < 1) return fail(c, DV_ERR_STATE, "no library or no resident patches");
    *device_ptr = c->d_record;
    *n_doubles = 3 + 4 * c->A;
    return DV_OK;
}

extern "C" int dv_step_keys(dv_ctx* c, int rank, int world, int signed_order, void** device_ptr, int* n_words) {
    if (!c || !device_ptr || !n_words) return DV_ERR_INVALID;
    if (!c->have_lib || c->A < 1) return fail(c, DV_ERR_STATE, "no library or no resident patches");
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(c, DV_ERR_INVALID, "dv_step_keys: rank %d of %d", rank, world);
    HIP_TRY(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(k_make_keys, dim3(1), dim3(64), 0, c->stream, c->d_record, c->d_keys, c->A, rank, world, signed_order ? 1 : 0);
    HIP_TRY(c, hipGetLastError());
    *device_ptr = c->d_keys;
    *n_words = c->A + kKeyWordsPerRank * world;
    return DV_OK;
}

extern "C" int dv_resolve_enqueue(dv_ctx* c) {
    if (!c) return DV_ERR_INVALID;
    if (!c->have_lib || !c->step_pending) return fail(c, DV_ERR_STATE, "no step to resolve");
    HIP_TRY(c, hipSetDevice(c->device));
    return enqueue_resolve(c);
}


extern "C" int dv_score_f32(dv_ctx* c, const float* patch, double* ssdbuf) {
    int rc = upload_patches_f32(c, patch, 1);
    if (rc) return rc;
    if (!ssdbuf) return fail(c, DV_ERR_INVALID, "ssdbuf is NULL");
    rc = launch_scoring(c);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(ssdbuf, c->d_fam, (size_t)c->cfg.F * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int64_t f = 0; f < c->cfg.F; ++f) ssdbuf[f] = -ssdbuf[f];
    c->step_pending = false;
    return DV_OK;
}

extern "C" int dv_score_u8(dv_ctx* c, const uint8_t* patch, double* ssdbuf) {
    int rc = upload_patches_u8(c, patch, 1);
    if (rc) return rc;
    if (!ssdbuf) return fail(c, DV_ERR_INVALID, "ssdbuf is NULL");
    rc = launch_scoring(c);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(ssdbuf, c->d_fam, (size_t)c->cfg.F * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    for (int64_t f = 0; f < c->cfg.F; ++f) ssdbuf[f] = -ssdbuf[f];
    c->step_pending = false;
    return DV_OK;
}

extern "C" int dv_score(dv_ctx* c, const uint8_t* patch, double* fambuf) {
    int rc = dv_upload_patches(c, patch, 1);
    if (rc) return rc;
    if (!fambuf) return fail(c, DV_ERR_INVALID, "fambuf is NULL");
    rc = launch_scoring(c);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpyAsync(fambuf, c->d_fam, (size_t)c->cfg.F * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->step_pending = false;
    return DV_OK;
}

// ------------------------------------------------------------------ error / coverage metrics

extern "C" int dv_set_training_path(dv_ctx* c, const double* xy, int64_t n) {
    if (!c) return DV_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
    if (c->d_path) { (void)hipFree(c->d_path); c->d_path = nullptr; }
    if (c->d_cover) { (void)hipFree(c->d_cover); c->d_cover = nullptr; }
    if (c->d_cover_slots) { (void)hipFree(c->d_cover_slots); c->d_cover_slots = nullptr; }
    c->n_cover_slots = 0;
    c->n_path = 0;
    c->err_enq = c->err_deq = 0;
    if (!xy || n < 1) return DV_OK;                              // detach
    HIP_TRY(c, hipMalloc(&c->d_path, (size_t)n * 2 * sizeof(double)));
    HIP_TRY(c, hipMalloc(&c->d_cover, (size_t)n));
    if (!c->d_errstate) {
        HIP_TRY(c, hipMalloc(&c->d_errstate, sizeof(PathErrState)));
        HIP_TRY(c, hipHostMalloc(&c->h_errout, kErrRing * sizeof(PathErrOut), hipHostMallocMapped));
        HIP_TRY(c, hipHostGetDevicePointer((void**)&c->d_errout, c->h_errout, 0));
    }
    memset(c->h_errout, 0, kErrRing * sizeof(PathErrOut));
    const PathErrState init{~0ull, 0u, 0u};
    HIP_TRY(c, hipMemcpyAsync(c->d_errstate, &init, sizeof init, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_path, xy, (size_t)n * 2 * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_cover, 0, (size_t)n, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->n_path = n;
    return DV_OK;
}

extern "C" int dv_path_error_enqueue(dv_ctx* c, double x, double y, double reach) {
    if (!c) return DV_ERR_INVALID;
    if (c->n_path < 1) return fail(c, DV_ERR_STATE, "no training path set (dv_set_training_path)");
    if (c->err_enq - c->err_deq >= (unsigned long long)kErrRing)
        return fail(c, DV_ERR_STATE, "%d path-error answers outstanding: collect them with dv_path_error_wait", kErrRing);
    HIP_TRY(c, hipSetDevice(c->device));
    // on its own stream: the answer is collected a step later (dv_path_error_wait), so it runs beside the next step's kernels
    // (the two forms are never in flight together -- they share d_errstate: a context's steps are either dv_agent_step's or not)
    if (c->err_on_main) { HIP_TRY(c, hipStreamSynchronize(c->stream)); c->err_on_main = false; }
    const PathErrArgs pe = path_err_args(c, x, y, reach);
    hipLaunchKernelGGL(k_path_error, dim3((unsigned)pe.nblk), dim3(256), 0, c->aux_stream, pe);
    HIP_TRY(c, hipGetLastError());
    return DV_OK;
}

extern "C" int dv_path_error_wait(dv_ctx* c, double* nearest) {
    if (!c || !nearest) return DV_ERR_INVALID;
    if (c->err_deq >= c->err_enq) return fail(c, DV_ERR_STATE, "no path-error answer outstanding");
    const unsigned long long seq = c->err_deq + 1;
    volatile const unsigned long long* word = &c->h_errout[seq % kErrRing].seq;
    bool seen = false;
    if (c->spin_wait) {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (!(seen = (*word == seq))) {
            if ((++spins & 1023u) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;
        }
    }
    if (!seen) {
        HIP_TRY(c, hipSetDevice(c->device));
        HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (*word != seq) return fail(c, DV_ERR_STATE, "the path-error answer never arrived");
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    *nearest = c->h_errout[seq % kErrRing].nearest;
    c->err_deq = seq;
    return DV_OK;
}

extern "C" int dv_path_coverage(dv_ctx* c, uint8_t* out, int64_t n) {
    if (!c || !out) return DV_ERR_INVALID;
    if (c->n_path < 1 || n != c->n_path) return fail(c, DV_ERR_STATE, "no training path of %lld points set", (long long)n);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
    HIP_TRY(c, hipMemcpyAsync(out, c->d_cover, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DV_OK;
}

extern "C" int dv_path_reset(dv_ctx* c) {
    if (!c) return DV_ERR_INVALID;
    if (c->n_path < 1) return DV_OK;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->aux_stream));
    c->err_deq = c->err_enq;                                     // answers of the run being abandoned are dropped
    HIP_TRY(c, hipMemsetAsync(c->d_cover, 0, (size_t)c->n_path, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DV_OK;
}

// update_error for the agents of an ensemble (include/dejavu.h): every agent has its own coverage marks on the device (a slot).
extern "C" int dv_path_slots(dv_ctx* c, int n_slots) {
    if (!c || n_slots < 0) return DV_ERR_INVALID;
    if (c->n_path < 1) return fail(c, DV_ERR_STATE, "no training path set (dv_set_training_path)");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_cover_slots) { (void)hipFree(c->d_cover_slots); c->d_cover_slots = nullptr; }
    c->n_cover_slots = 0;
    if (n_slots == 0) return DV_OK;
    HIP_TRY(c, hipMalloc(&c->d_cover_slots, (size_t)n_slots * (size_t)c->n_path));
    HIP_TRY(c, hipMemsetAsync(c->d_cover_slots, 0, (size_t)n_slots * (size_t)c->n_path, c->stream));
    if (!c->d_minkeys) {
        HIP_TRY(c, hipMalloc(&c->d_minkeys, kPathBatch * sizeof(unsigned long long)));
        HIP_TRY(c, hipHostMalloc(&c->h_minkeys, kPathBatch * sizeof(unsigned long long), hipHostMallocDefault));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->n_cover_slots = n_slots;
    return DV_OK;
}

extern "C" int dv_path_error_batch(dv_ctx* c, const int32_t* slots, const double* x, const double* y, int n, double reach, double* nearest) {
    if (!c || !slots || !x || !y || !nearest || n < 0) return DV_ERR_INVALID;
    if (c->n_path < 1 || c->n_cover_slots < 1) return fail(c, DV_ERR_STATE, "no coverage slots (dv_set_training_path, dv_path_slots)");
    for (int i = 0; i < n; ++i)
        if (slots[i] < 0 || slots[i] >= c->n_cover_slots) return fail(c, DV_ERR_INVALID, "dv_path_error_batch: slot %d of %d", slots[i], c->n_cover_slots);
    HIP_TRY(c, hipSetDevice(c->device));
    long long nb = (c->n_path + 1023) / 1024;                         // 1024 points per block of 256 threads (path_err_args)
    if (nb > 256) nb = 256;
    for (int first = 0; first < n; first += kPathBatch) {
        const int cnt = n - first < kPathBatch ? n - first : kPathBatch;
        PathBatchArgs pa{};
        pa.xy = c->d_path; pa.n = (long long)c->n_path; pa.reach = reach; pa.cover = c->d_cover_slots; pa.cover_stride = (long long)c->n_path;
        pa.minkey = c->d_minkeys;
        for (int i = 0; i < cnt; ++i) { pa.x[i] = x[first + i]; pa.y[i] = y[first + i]; pa.slot[i] = slots[first + i]; }
        HIP_TRY(c, hipMemsetAsync(c->d_minkeys, 0xff, kPathBatch * sizeof(unsigned long long), c->stream));
        hipLaunchKernelGGL(k_path_error_batch, dim3((unsigned)nb, (unsigned)cnt), dim3(256), 0, c->stream, pa);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(c->h_minkeys, c->d_minkeys, (size_t)cnt * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (int i = 0; i < cnt; ++i) memcpy(&nearest[first + i], &c->h_minkeys[i], sizeof(double));
    }
    return DV_OK;
}

extern "C" int dv_path_coverage_slot(dv_ctx* c, int slot, uint8_t* out, int64_t n) {
    if (!c || !out) return DV_ERR_INVALID;
    if (c->n_path < 1 || n != c->n_path || slot < 0 || slot >= c->n_cover_slots) return fail(c, DV_ERR_STATE, "no coverage slot %d of %lld points", slot, (long long)n);
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(out, c->d_cover_slots + (size_t)slot * (size_t)c->n_path, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DV_OK;
}

extern "C" int dv_path_reset_slot(dv_ctx* c, int slot) {
    if (!c) return DV_ERR_INVALID;
    if (c->n_cover_slots < 1) return DV_OK;
    if (slot >= c->n_cover_slots) return fail(c, DV_ERR_INVALID, "dv_path_reset_slot: slot %d of %d", slot, c->n_cover_slots);
    HIP_TRY(c, hipSetDevice(c->device));
    if (slot < 0) HIP_TRY(c, hipMemsetAsync(c->d_cover_slots, 0, (size_t)c->n_cover_slots * (size_t)c->n_path, c->stream));
    else HIP_TRY(c, hipMemsetAsync(c->d_cover_slots + (size_t)slot * (size_t)c->n_path, 0, (size_t)c->n_path, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return DV_OK;
}

// ------------------------------------------------------------------ measurement
#ifdef DEJAVU_STAMPS
extern "C" int dv_debug_stamps(dv_ctx* c, unsigned long long* out) {       // diagnostic builds only (tools/exp/stamps.py)
    if (!c || !out) return DV_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dv_stamps), 256 * 8 * sizeof(unsigned long long)));
    return DV_OK;
}
#endif

extern "C" int dv_timer_start(dv_ctx* c) {
    if (!c) return DV_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventRecord(c->t0, c->stream));
    return DV_OK;
}

extern "C" int dv_timer_stop(dv_ctx* c, float* ms) {
    if (!c || !ms) return DV_ERR_INVALID;
    HIP_TRY(c, hipEventRecord(c->t1, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->t1));
    HIP_TRY(c, hipEventElapsedTime(ms, c->t0, c->t1));
    return DV_OK;
}

extern "C" int dv_workgroup_shape(dv_ctx* c, int A, int* shape) {
    if (!c || !shape) return DV_ERR_INVALID;
    if (A < 1 || A > kMaxHeadings) return fail(c, DV_ERR_INVALID, "n_headings %d outside [1, %d]", A, kMaxHeadings);
    const int apad = A <= 8 ? 8 : (A <= 16 ? 16 : (A <= 32 ? 32 : 64));
    *shape = (c->have_lib && c->metric == 0 && !c->cfg.generic) ? (c->shape_env ? c->shape_env : c->tuned_shape[apad_class(apad)]) : 0;
    return DV_OK;
}

extern "C" int dv_profile_kernel(dv_ctx* c, int enable) {
    if (!c) return DV_ERR_INVALID;
    c->profile = enable > 0 ? enable : 0;
    c->profile_count = 0;
    c->pev_used = 0;
    return DV_OK;
}

extern "C" int dv_profile_read(dv_ctx* c, double* total_ms, int64_t* n) {
    if (!c || !total_ms || !n) return DV_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    double sum = 0.0;
    for (size_t i = 0; i + 1 < c->pev_used; i += 2) {
        float ms = 0.f;
        HIP_TRY(c, hipEventElapsedTime(&ms, c->pev[i], c->pev[i + 1]));
        sum += ms;
    }
    *total_ms = sum;
    *n = (int64_t)(c->pev_used / 2);
    c->pev_used = 0;
    return DV_OK;
}

extern "C" int dv_stream_read_gbps(dv_ctx* c, int64_t n_bytes, int iters, double* gbps) {
    if (!c || !gbps || n_bytes < 16 || iters < 1) return DV_ERR_INVALID;
    HIP_TRY(c, hipSetDevice(c->device));
    uint4* buf = nullptr;
    unsigned* sink = nullptr;
    const long long n16 = n_bytes / 16;
    HIP_TRY(c, hipMalloc(&buf, (size_t)n16 * 16));
    hipError_t e = hipMalloc(&sink, 16);
    if (e == hipSuccess) e = hipMemsetAsync(buf, 0x5a, (size_t)n16 * 16, c->stream);
    float ms = 0.f;
    if (e == hipSuccess) {
        const dim3 grid(256), block(512);
        const size_t lds = (size_t)8 * kStreamRows * 1024;
        (void)hipFuncSetAttribute((const void*)k_stream_read, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(k_stream_read, grid, block, lds, c->stream, buf, n16, sink);   // warm-up
        e = hipEventRecord(c->t0, c->stream);
        for (int i = 0; i < iters && e == hipSuccess; ++i) {
            hipLaunchKernelGGL(k_stream_read, grid, block, lds, c->stream, buf, n16, sink);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(c->t1, c->stream);
        if (e == hipSuccess) e = hipEventSynchronize(c->t1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, c->t0, c->t1);
    }
    (void)hipFree(buf);
    if (sink) (void)hipFree(sink);
    if (e != hipSuccess) return fail(c, DV_ERR_HIP, "stream_read: %s", hipGetErrorString(e));
    *gbps = (double)n16 * 16.0 * iters / (ms * 1e-3) / 1e9;
    return DV_OK;
}

#include "dejavu_group.inl"     // dv_group_*: one process, several devices -- host logic above the C ABI
