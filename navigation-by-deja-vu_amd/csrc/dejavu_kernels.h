// dejavu_kernels.h -- gfx950 (CDNA4) device code of the scene-familiarity engine.
//
// Device layout of the stored-view library ("tiles"), owned by this library:
//
//   tiles[g][plane][q][lane] : uint4 (16 B)
//     g     = view group, 64 consecutive views           (g = f / 64, lane = f % 64)
//     plane = one byte per pixel: one-hot saturation planes, then the value plane
//     q     = chunk of 16 consecutive pixels of the flattened h*w sensor (zero padded)
//
// One wave64 handles one view group: lane <-> view, so a wave-wide dwordx4 load is 1 KiB
// contiguous, every lane accumulates its own view's sums with v_sad_u8 (4 pixels per
// lane-op) and no cross-lane reduction is needed until the final max over views.  The patch
// operand of each v_sad_u8 is the same for all 64 lanes, so it is read with scalar loads
// (s_load_dwordx8/16 -> SGPR operand): the patches never occupy VGPRs or LDS bandwidth.
//
// Hue-aware saturation term of the reference (navsim/util.pyx:48-56):
//     hs = (H_s == H_f) ? |S_s - S_f| : S_s + S_f
// equals the L1 distance between the one-hot vectors S*e_H, so with the library's hue set
// K = {hues with S > 0} stored as |K| planes  S_k = (H == k ? S : 0):
//     hs = sum_k |S_s,k - S_f,k|  +  (H_s not in K ? S_s : 0)
// which is |K| v_sad_u8 per 4 pixels plus a per-heading constant.  Exact for all inputs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#pragma clang fp contract(off)

namespace dv {

// Diagnostic builds only (tools/exp/stamps.py compiles a copy with -DDEJAVU_STAMPS): wall-clock stamps (100 MHz counter)
// of the matrix-core kernel's phases per workgroup, in a buffer nothing else reads.  The product build has no stamps.
#ifdef DEJAVU_STAMPS
#ifdef DEJAVU_EXP_FIN
#define DV_EXP_FIN_ON 1
#else
#define DV_EXP_FIN_ON 0
#endif
__device__ unsigned long long g_dv_stamps[256 * 8];
// (slots 6 and 7: the shader-clock counter beside stamps 1 and 2 -- the clock the CU held during the first item's loop)
#define DV_STAMP(i) do { if (threadIdx.x == 0) { g_dv_stamps[(blockIdx.x & 255) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); \
    if (((i) == 1 || (i) == 2) && !DV_EXP_FIN_ON) g_dv_stamps[(blockIdx.x & 255) * 8 + 5 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define DV_STAMP(i) do { } while (0)
#endif

constexpr int kMaxHeadings = 64;
constexpr int kMaxHues = 4;
constexpr int kCandCap = 4096;

struct LibCfg {
    long long F;        // local views
    long long Fpad;     // padded to a multiple of 64
    long long first;    // global index of local view 0
    long long gstride;  // 16-byte units between consecutive view groups in the tile array (>= npl*Q*64: padded so that
                        // concurrently streamed groups do not sit a power of two apart)
    int P;              // h*w
    int Q;              // ceil(P/16)
    int npl;            // planes stored per pixel
    int nhs;            // one-hot saturation planes (or 2 = H,S when generic)
    int hasv;           // value plane present
    int generic;        // H and S kept as planes (more than kMaxHues hues)
    int signed_s;       // two hues and every library S <= 127: ONE saturation plane holding 128 + (S of hue0) - (S of hue1)
    int synth_full_s;   // the synthetic generators (dv_generate_library_ex, dv_generate_patches) draw S from 0..127 instead of {0, 127}
    unsigned char hues[kMaxHues];
    double cw;          // chem_weight
    double whs;         // 0.5 * cw          (util.pyx:59,68)
    double wv;          // 1 - cw            (util.pyx:69)
};

typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
// Per-heading constants of a patch preparation (PrepAcc, k_patch_prep): a heading's sum is kept as kAccWays partial sums, each in
// a cache line of its own (kAccStride ints apart) -- the blocks of a heading add to them with atomics, and atomics to ONE line
// serialise in L2.
constexpr int kAccStride = 32, kAccWays = 4;
__device__ __forceinline__ int acc_sum(const int* __restrict__ p, int a) {
    int t = 0;
#pragma unroll
    for (int w = 0; w < kAccWays; ++w) t += p[(w * kMaxHeadings + a) * kAccStride];
    return t;
}
struct StepState {                       // zeroed by the scoring epilogue (k_combine / k_exact_all) of every step
    unsigned long long amax[kMaxHeadings];      // ordered key of max_f fam[a][f]
    unsigned long long aview[kMaxHeadings];     // ~f of the first view attaining it (0 = none)
    unsigned long long ncand;                   // candidates found (may exceed kCandCap)
    unsigned done;                              // blocks of k_tail / k_finish that have finished
    unsigned ntmp;                              // k_finish: extra candidates in flight (zero between steps)
};

__device__ __forceinline__ void reset_step_state(StepState* st, int tid, int n_agents) {
    for (int ag = 0; ag < n_agents; ++ag) {
        if (tid < kMaxHeadings) st[ag].aview[tid] = 0;
        if (tid == 0) { st[ag].ncand = 0; st[ag].done = 0; }
    }
}

// ------------------------------------------------------------------ helpers
__device__ __forceinline__ unsigned long long ordered_key(double d) {
    unsigned long long b = (unsigned long long)__double_as_longlong(d);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_to_double(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)b);
}
__host__ __device__ __forceinline__ unsigned long long splitmix64(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// Same bit fields as navsim_amd/synth.py:hsv_from_words.
__device__ __forceinline__ void synth_hsv(unsigned long long z, unsigned& H, unsigned& S, unsigned& V, int full_s = 0) {
    const unsigned lvl = (unsigned)(((z & 0xFFFFull) * 5ull) >> 16);
    const unsigned levels[5] = {0u, 63u, 127u, 191u, 255u};
    V = levels[lvl];
    H = (unsigned)((z >> 16) & 1ull) * 127u;
    S = full_s ? (unsigned)((z >> 17) & 0x7Full) : (unsigned)((z >> 17) & 1ull) * 127u;
}
// Byte stored in plane `pl` for a pixel (H,S,V).
// Signed-saturation plane: with two hues, |S_s,0 - S_f,0| + |S_s,1 - S_f,1| = |x_s - x_f| for x = S_0 - S_1 (at most one of
// the two is nonzero per pixel), so ONE byte plane 128 + x replaces two whenever |x| <= 127.  Library pixels satisfy that
// by construction (checked at ingest); a patch pixel with S > 127 is clamped to +-127 and the excess S - 127 goes into
// the per-heading constant, which is exact because |+-(127 + r) - x_f| = |+-127 - x_f| + r for every |x_f| <= 127.
__device__ __forceinline__ unsigned plane_byte(const LibCfg& c, int pl, unsigned H, unsigned S, unsigned V) {
    if (c.generic) {
        if (pl < c.nhs) return pl == 0 ? H : S;
        return V;
    }
    if (c.signed_s && pl == 0) {
        const unsigned s = S > 127u ? 127u : S;
        return H == c.hues[0] ? 128u + s : (H == c.hues[1] ? 128u - s : 128u);
    }
    if (pl < c.nhs) return (H == c.hues[pl]) ? S : 0u;
    return V;
}
// Reference's per-pixel term in its exact operation order (navsim/util.pyx:48-72).
__device__ __forceinline__ double px_term(int hs, int dv, double cw, double wv) {
    double t = (double)hs;
    t *= 0.5;
    t *= cw;
    t += wv * (double)dv;
    t /= 255.;
    return t;
}
// Integer hue/saturation term and |dV| of one pixel from the stored planes of a view
// (bytes lib[pl]) and the raw patch pixel (H,S,V).
__device__ __forceinline__ void px_ints(const LibCfg& c, const unsigned* lib, unsigned H, unsigned S, unsigned V,
                                        int& hs, int& dv) {
    hs = 0;
    dv = 0;
    if (c.cw > 0.0) {
        if (c.generic) {
            const int lh = (int)lib[0], ls = (int)lib[1];
            hs = ((int)H == lh) ? abs((int)S - ls) : (int)S + ls;
        } else if (c.signed_s) {
            const int x = (int)lib[0] - 128;                  // +S of hue 0, -S of hue 1
            const int f0 = x > 0 ? x : 0, f1 = x < 0 ? -x : 0;
            const int s0 = H == c.hues[0] ? (int)S : 0, s1 = H == c.hues[1] ? (int)S : 0;
            hs = abs(s0 - f0) + abs(s1 - f1);
            if (H != c.hues[0] && H != c.hues[1]) hs += (int)S;
        } else {
            bool in_set = false;
            for (int k = 0; k < c.nhs; ++k) {
                const bool mine = (H == c.hues[k]);
                in_set |= mine;
                hs += abs((mine ? (int)S : 0) - (int)lib[k]);
            }
            if (!in_set) hs += (int)S;
        }
    }
    if (c.hasv) dv = abs((int)V - (int)lib[c.nhs]);
}

// ------------------------------------------------------------------ library construction
// Marks every hue that occurs with S > 0 in the raw library (uint8[n_px][3]); bitmap[8] = largest S.
__global__ void k_hue_scan(const unsigned char* __restrict__ raw, long long n_px, unsigned* __restrict__ bitmap) {
    __shared__ unsigned local[9];
    if (threadIdx.x < 9) local[threadIdx.x] = 0;
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned smax = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_px; i += stride) {
        const unsigned H = raw[i * 3 + 0], S = raw[i * 3 + 1];
        if (S > 0) atomicOr(&local[H >> 5], 1u << (H & 31));
        smax = S > smax ? S : smax;
    }
    atomicMax(&local[8], smax);
    __syncthreads();
    if (threadIdx.x < 8 && local[threadIdx.x]) atomicOr(&bitmap[threadIdx.x], local[threadIdx.x]);
    if (threadIdx.x == 8) atomicMax(&bitmap[8], local[8]);
}

// raw uint8[F][P][3] -> tiles.  One thread per 16-byte chunk (g, plane, q, lane).
__global__ void k_retile(const unsigned char* __restrict__ raw, uint4* __restrict__ tiles, LibCfg c) {
    const long long total = (c.Fpad / 64) * (long long)c.npl * c.Q * 64;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    long long r = t >> 6;
    const int q = (int)(r % c.Q); r /= c.Q;
    const int pl = (int)(r % c.npl);
    const long long g = r / c.npl;
    const long long f = g * 64 + lane;
    unsigned w[4] = {0, 0, 0, 0};
    if (f < c.F) {
        const unsigned char* v = raw + f * (long long)c.P * 3;
        for (int i = 0; i < 16; ++i) {
            const int px = q * 16 + i;
            if (px < c.P) {
                const unsigned b = plane_byte(c, pl, v[px * 3 + 0], v[px * 3 + 1], v[px * 3 + 2]);
                w[i >> 2] |= b << (8 * (i & 3));
            }
        }
    }
    tiles[g * c.gstride + ((long long)pl * c.Q + q) * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
}

// Appended views: raw uint8[n][P][3] holds local views [f0, f0 + n) of the grown library.  One thread per 16-byte chunk
// of the view groups from f0 / 64 on; entries of views below f0 (already in place) are left alone.
__global__ void k_retile_append(const unsigned char* __restrict__ raw, uint4* __restrict__ tiles, LibCfg c, long long f0) {
    const long long g0 = f0 / 64;
    const long long total = (c.Fpad / 64 - g0) * (long long)c.npl * c.Q * 64;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    long long r = t >> 6;
    const int q = (int)(r % c.Q); r /= c.Q;
    const int pl = (int)(r % c.npl);
    const long long g = g0 + r / c.npl;
    const long long f = g * 64 + lane;
    if (f < f0) return;
    unsigned w[4] = {0, 0, 0, 0};
    if (f < c.F) {
        const unsigned char* v = raw + (f - f0) * (long long)c.P * 3;
        for (int i = 0; i < 16; ++i) {
            const int px = q * 16 + i;
            if (px < c.P) {
                const unsigned b = plane_byte(c, pl, v[px * 3 + 0], v[px * 3 + 1], v[px * 3 + 2]);
                w[i >> 2] |= b << (8 * (i & 3));
            }
        }
    }
    tiles[g * c.gstride + ((long long)pl * c.Q + q) * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
}

// Synthetic library straight into tiles: view f, pixel p <- splitmix64((first+f)*P + p + seed*GOLDEN).
__global__ void k_generate_tiles(uint4* __restrict__ tiles, LibCfg c, unsigned long long seed) {
    const long long total = (c.Fpad / 64) * (long long)c.npl * c.Q * 64;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    long long r = t >> 6;
    const int q = (int)(r % c.Q); r /= c.Q;
    const int pl = (int)(r % c.npl);
    const long long g = r / c.npl;
    const long long f = g * 64 + lane;
    unsigned w[4] = {0, 0, 0, 0};
    if (f < c.F) {
        const unsigned long long base = (unsigned long long)(c.first + f) * (unsigned long long)c.P +
                                        seed * 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < 16; ++i) {
            const int px = q * 16 + i;
            if (px < c.P) {
                unsigned H, S, V;
                synth_hsv(splitmix64(base + (unsigned long long)px), H, S, V, c.synth_full_s);
                w[i >> 2] |= plane_byte(c, pl, H, S, V) << (8 * (i & 3));
            }
        }
    }
    tiles[g * c.gstride + ((long long)pl * c.Q + q) * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
}


// tiles -> planes uint8[n][npl][P] for local views [v0, v0+n)   (layout read-back)
__global__ void k_read_planes(const uint4* __restrict__ tiles, unsigned char* __restrict__ out, LibCfg c,
                              long long v0, long long n) {
    const long long total = n * c.npl * (long long)c.P;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int px = (int)(t % c.P);
    long long r = t / c.P;
    const int pl = (int)(r % c.npl);
    const long long f = v0 + r / c.npl;
    const long long idx = (f >> 6) * c.gstride + ((long long)pl * c.Q + (px >> 4)) * 64 + (f & 63);
    const unsigned char* b = reinterpret_cast<const unsigned char*>(tiles + idx);
    out[t] = b[px & 15];
}

// ------------------------------------------------------------------ the scoring kernel
// Work item = (pixel chunk c, view group g): 64 views x the chunk's pixels x all planes, scored by
// ONE wave (lane <-> view).  Items are numbered chunk-major (item = c*G + g) and a fixed grid of
// single-wave workgroups walks them with a grid stride, so
//   * every CU gets the same number of items to within one (HBM delivers ~1/256 of the chip's
//     bandwidth to each CU, ~25 GB/s: a CU holding 7 workgroups where others hold 6 finishes 1/6
//     later and sets the kernel time -- measured with per-wave stamps in exp/sad_trace.hip);
//   * the grid never exceeds what is resident at once (no second, almost empty round);
//   * waves that run at the same time work on the same pixel chunk, so they read the same patch
//     dwords and the scalar cache serves most s_loads (42 % hits + 42 % hit-on-miss measured).
// The kernel's only output is its raw integer sums: part[c][s][a][f] (u32), plain coalesced
// stores; k_combine adds the chunks up.  Loads are non-temporal: each byte is used once per step
// (default-policy loads measured 13 % slower on the same structure).
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 load_tile_nt(const uint4* p) {
    const v4u_t t = __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(p));   // read once per step
    return make_uint4(t.x, t.y, t.z, t.w);
}

template <bool NT>
__device__ __forceinline__ uint4 load_tile(const uint4* p) {
    if (NT) return load_tile_nt(p);
    return *p;                                  // several waves of the workgroup read this tile: keep it cached
}

// One-hot layout.  NHS saturation planes + optional value plane; APAD headings per pass.
// Scores headings [a_off, a_off+APAD) of the ATOT resident ones (ATOT == APAD except for the two 32-wide passes
// that cover up to 64 headings).
template <int NHS, int HASV, int APAD, int ATOT, int NW, int HW>
__global__ void __launch_bounds__(64 * NW * HW) __attribute__((amdgpu_num_sgpr(96)))
k_sad_tiles(const uint4* __restrict__ tiles, const unsigned* __restrict__ prep, unsigned* __restrict__ part, LibCfg c,
            int nchunk, int a_off_arg) {
    extern __shared__ unsigned red_all[];      // [HW][NSUM*APAD][64], only when NW > 1
    constexpr int apad_total = ATOT;
    constexpr int NPL = NHS + HASV;
    constexpr int NSUM = (NHS > 0 ? 1 : 0) + HASV;
    // Register ring depth: chunk q+PF is in flight while q is scored.  Few planes / few headings leave VGPRs for a
    // deeper ring (more bytes in flight per wave); with 2 or 3 planes PF = 1 measured best (A/B, tools/ab_lib.sh).
    constexpr int PF = (APAD <= 16 && NPL == 1) ? 3 : 1;
    const int lane = threadIdx.x & 63;
    constexpr int nw = NW;
    // Workgroup = HW "heading ways" x NW waves.  The NW waves of a way share the item's 16-pixel steps (below); the HW
    // ways score the same tiles against different APAD-wide slices of the resident headings, so one trip of the
    // library through HBM serves HW*APAD headings (the ways run in step and meet in L1/L2).
    const int wid = (NW * HW == 1) ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave = NW == 1 ? 0 : (HW == 1 ? wid : wid % NW);
    const int hway = HW == 1 ? 0 : (NW == 1 ? wid : wid / NW);
    const int a_off = ((ATOT == APAD) ? 0 : a_off_arg) + hway * APAD;
    unsigned* red = red_all + hway * (NSUM * APAD * 64);
    const int Q = c.Q;
    const long long G = c.Fpad / 64;
    const long long n_items = G * nchunk;

    // The nw waves of a workgroup share an item and take its 16-pixel steps round-robin (step k of wave w is
    // q0 + w + k*nw): they read neighbouring 1 KB tiles and the same patch dwords at the same time.
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ch = (int)(item / G);
        const long long g = item - (long long)ch * G;
        const int q0 = (int)(((long long)ch * Q) / nchunk), q1 = (int)(((long long)(ch + 1) * Q) / nchunk);
        const int nk = (q1 - q0 - wave + nw - 1) / nw;          // steps of this wave (<= 0: none)
        const int qw = q0 + wave;
        const uint4* base = tiles + g * c.gstride + lane;

        unsigned acc_hs[NHS > 0 ? APAD : 1];
        unsigned acc_v[HASV ? APAD : 1];
#pragma unroll
        for (int a = 0; a < (NHS > 0 ? APAD : 1); ++a) acc_hs[a] = 0;
#pragma unroll
        for (int a = 0; a < (HASV ? APAD : 1); ++a) acc_v[a] = 0;

        if (nk > 0) {
            uint4 ring[PF + 1][NPL];
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                const int qq = qw + (s < nk ? s : nk - 1) * nw;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) ring[s][pl] = load_tile<HW == 1>(&base[(long long)(pl * Q + qq) * 64]);
            }
            for (int k = 0; k < nk; k += PF + 1) {
#pragma unroll
                for (int s = 0; s <= PF; ++s) {
                    const int kc = k + s;
                    const int qc = qw + kc * nw;
                    const int qn = qw + ((kc + PF < nk) ? kc + PF : nk - 1) * nw;
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl)
                        ring[(s + PF) % (PF + 1)][pl] = load_tile<HW == 1>(&base[(long long)(pl * Q + qn) * 64]);
                    if (kc < nk) {
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl) {
                            const unsigned* pp = prep + ((long long)(pl * Q + qc) * 4) * apad_total + a_off;   // wave-uniform -> s_load
                            const unsigned lw[4] = {ring[s][pl].x, ring[s][pl].y, ring[s][pl].z, ring[s][pl].w};
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
#pragma unroll
                                for (int a = 0; a < APAD; ++a) {
                                    if (pl < NHS) acc_hs[a] = __builtin_amdgcn_sad_u8(lw[j], pp[j * apad_total + a], acc_hs[a]);
                                    else acc_v[a] = __builtin_amdgcn_sad_u8(lw[j], pp[j * apad_total + a], acc_v[a]);
                                }
                            }
                        }
                    }
                }
            }
        }
        unsigned* dst = part + ((long long)ch * NSUM * apad_total + a_off) * c.Fpad + g * 64 + lane;
        if (nw == 1) {
            if (NHS > 0) {
#pragma unroll
                for (int a = 0; a < APAD; ++a) dst[(long long)a * c.Fpad] = acc_hs[a];
            }
            if (HASV) {
#pragma unroll
                for (int a = 0; a < APAD; ++a) dst[(long long)((NHS > 0 ? apad_total : 0) + a) * c.Fpad] = acc_v[a];
            }
            continue;
        }
        // the workgroup's waves add their sums up in LDS and share the stores: 1/nw of the partial-sum traffic
        if (wave == 0) {
            if (NHS > 0) {
#pragma unroll
                for (int a = 0; a < APAD; ++a) red[a * 64 + lane] = acc_hs[a];
            }
            if (HASV) {
#pragma unroll
                for (int a = 0; a < APAD; ++a) red[((NHS > 0 ? APAD : 0) + a) * 64 + lane] = acc_v[a];
            }
        }
        __syncthreads();
        if (wave != 0) {
            if (NHS > 0) {
#pragma unroll
                for (int a = 0; a < APAD; ++a) atomicAdd(&red[a * 64 + lane], acc_hs[a]);
            }
            if (HASV) {
#pragma unroll
                for (int a = 0; a < APAD; ++a) atomicAdd(&red[((NHS > 0 ? APAD : 0) + a) * 64 + lane], acc_v[a]);
            }
        }
        __syncthreads();
        for (int r = wave; r < NSUM * APAD; r += nw) {
            const int row = (NHS > 0 && r >= APAD) ? apad_total + (r - APAD) : r;
            dst[(long long)row * c.Fpad] = red[r * 64 + lane];
        }
        __syncthreads();
    }
}

// Packed variant for libraries that need both sums (0 < chem_weight < 1): one 32-bit accumulator per heading
// carries the value-plane SAD in its low half (v_sad_u8) and the saturation-plane SAD in its high half
// (v_sad_hi_u8: D = (SAD << 16) + S2), so a wave holds APAD accumulators instead of 2*APAD -- more waves per SIMD, or
// a deeper load ring, or 64 headings in one pass.  A half holds 16 steps of 16 pixels (256 * 255 < 2^16; with several
// saturation planes a pixel contributes up to 510, hence 8 steps); after that many steps the wave adds both halves
// into the workgroup's LDS accumulators, which the four waves of the workgroup (sharing the item's steps round-robin,
// as in k_sad_tiles with NW = 4) then store as one set of partial sums.
template <int NHS, int APAD, int PF>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96)))
k_sad_packed(const uint4* __restrict__ tiles, const unsigned* __restrict__ prep, unsigned* __restrict__ part, LibCfg c,
             int nchunk) {
    extern __shared__ unsigned red[];          // [2*APAD][64]: rows 0..APAD-1 saturation sums, then value sums
    constexpr int NPL = NHS + 1;
    constexpr int NW = 4;
    constexpr int FLUSH = (NHS > 1 ? 8 : 16) / (PF + 1);     // ring turns between flushes
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Q = c.Q;
    const long long G = c.Fpad / 64;
    const long long n_items = G * nchunk;

    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ch = (int)(item / G);
        const long long g = item - (long long)ch * G;
        const int q0 = (int)(((long long)ch * Q) / nchunk), q1 = (int)(((long long)(ch + 1) * Q) / nchunk);
        const int nk = (q1 - q0 - wave + NW - 1) / NW;
        const int qw = q0 + wave;
        const uint4* base = tiles + g * c.gstride + lane;

        for (int r = wave; r < 2 * APAD; r += NW) red[r * 64 + lane] = 0;
        __syncthreads();

        if (nk > 0) {
            uint4 ring[PF + 1][NPL];
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                const int qq = qw + (s < nk ? s : nk - 1) * NW;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) ring[s][pl] = load_tile_nt(&base[(long long)(pl * Q + qq) * 64]);
            }
            // blocks of FLUSH ring turns: the accumulators start at zero in each block and are flushed after it
            for (int kb = 0; kb < nk; kb += FLUSH * (PF + 1)) {
                unsigned acc[APAD];
#pragma unroll
                for (int a = 0; a < APAD; ++a) acc[a] = 0;
                const int kend = (kb + FLUSH * (PF + 1) < nk) ? kb + FLUSH * (PF + 1) : nk;
                for (int k = kb; k < kend; k += PF + 1) {
#pragma unroll
                    for (int s = 0; s <= PF; ++s) {
                        const int kc = k + s;
                        const int qc = qw + kc * NW;
                        const int qn = qw + ((kc + PF < nk) ? kc + PF : nk - 1) * NW;
#pragma unroll
                        for (int pl = 0; pl < NPL; ++pl)
                            ring[(s + PF) % (PF + 1)][pl] = load_tile_nt(&base[(long long)(pl * Q + qn) * 64]);
                        if (kc < nk) {
#pragma unroll
                            for (int pl = 0; pl < NPL; ++pl) {
                                const unsigned* pp = prep + ((long long)(pl * Q + qc) * 4) * APAD;   // wave-uniform -> s_load
                                const unsigned lw[4] = {ring[s][pl].x, ring[s][pl].y, ring[s][pl].z, ring[s][pl].w};
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
#pragma unroll
                                    for (int a = 0; a < APAD; ++a) {
                                        if (pl < NHS) acc[a] = __builtin_amdgcn_sad_hi_u8(lw[j], pp[j * APAD + a], acc[a]);
                                        else acc[a] = __builtin_amdgcn_sad_u8(lw[j], pp[j * APAD + a], acc[a]);
                                    }
                                }
                            }
                        }
                    }
                }
#pragma unroll
                for (int a = 0; a < APAD; ++a) {
                    atomicAdd(&red[a * 64 + lane], acc[a] >> 16);
                    atomicAdd(&red[(APAD + a) * 64 + lane], acc[a] & 0xffffu);
                }
            }
        }
        __syncthreads();
        unsigned* dst = part + ((long long)ch * 2 * APAD) * c.Fpad + g * 64 + lane;
        for (int r = wave; r < 2 * APAD; r += NW) dst[(long long)r * c.Fpad] = red[r * 64 + lane];
        __syncthreads();
    }
}

// Generic-hue layout (planes H,S[,V]): per-byte hue compare done with bit tricks.
template <int HAS_HS, int HASV, int APAD, int ATOT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(96)))
k_sad_generic(const uint4* __restrict__ tiles, const unsigned* __restrict__ prep, unsigned* __restrict__ part, LibCfg c,
              int nchunk, int a_off_arg) {
    constexpr int apad_total = ATOT;
    const int a_off = (ATOT == APAD) ? 0 : a_off_arg;
    constexpr int NPL = (HAS_HS ? 2 : 0) + HASV;
    constexpr int NSUM = HAS_HS + HASV;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const int Q = c.Q;
    const long long G = c.Fpad / 64;
    const long long n_items = G * nchunk;
    const long long stride = (long long)gridDim.x * nw;
  for (long long item = (long long)blockIdx.x * nw + wave; item < n_items; item += stride) {
    const int ch = (int)(item / G);
    const long long g = item - (long long)ch * G;
    const int q0 = (int)(((long long)ch * Q) / nchunk), q1 = (int)(((long long)(ch + 1) * Q) / nchunk);
    const uint4* base = tiles + g * c.gstride + lane;

    unsigned acc_hs[HAS_HS ? APAD : 1];
    unsigned acc_v[HASV ? APAD : 1];
#pragma unroll
    for (int a = 0; a < (HAS_HS ? APAD : 1); ++a) acc_hs[a] = 0;
#pragma unroll
    for (int a = 0; a < (HASV ? APAD : 1); ++a) acc_v[a] = 0;

    for (int q = q0; q < q1; ++q) {
        uint4 L[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) L[pl] = load_tile_nt(&base[(long long)(pl * Q + q) * 64]);
        if constexpr (HAS_HS != 0) {
            const unsigned* ph = prep + ((long long)(0 * Q + q) * 4) * apad_total + a_off;
            const unsigned* ps = prep + ((long long)(1 * Q + q) * 4) * apad_total + a_off;
            const unsigned lh[4] = {L[0].x, L[0].y, L[0].z, L[0].w};
            const unsigned ls[4] = {L[1].x, L[1].y, L[1].z, L[1].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int a = 0; a < APAD; ++a) {
                    const unsigned x = lh[j] ^ ph[j * apad_total + a];
                    unsigned t = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;  // 0x80 where hue differs
                    const unsigned ne = (t - (t >> 7)) | t;                                  // 0xFF where hue differs
                    const unsigned sp = ps[j * apad_total + a];
                    unsigned s = acc_hs[a];
                    s = __builtin_amdgcn_sad_u8(sp & ~ne, ls[j] & ~ne, s);   // same hue: |S_s - S_f|
                    s = __builtin_amdgcn_sad_u8(sp & ne, 0u, s);             // different hue: S_s + S_f
                    s = __builtin_amdgcn_sad_u8(ls[j] & ne, 0u, s);
                    acc_hs[a] = s;
                }
            }
        }
        if constexpr (HASV != 0) {
            constexpr int VP = HAS_HS ? 2 : 0;
            const unsigned* pv = prep + ((long long)(VP * Q + q) * 4) * apad_total + a_off;
            const unsigned lv[4] = {L[VP].x, L[VP].y, L[VP].z, L[VP].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int a = 0; a < APAD; ++a) acc_v[a] = __builtin_amdgcn_sad_u8(lv[j], pv[j * apad_total + a], acc_v[a]);
            }
        }
    }
    unsigned* dst = part + ((long long)ch * NSUM * apad_total + a_off) * c.Fpad + g * 64 + lane;
    if (HAS_HS) {
#pragma unroll
        for (int a = 0; a < APAD; ++a) dst[(long long)a * c.Fpad] = acc_hs[a];
    }
    if (HASV) {
#pragma unroll
        for (int a = 0; a < APAD; ++a) dst[(long long)((HAS_HS ? apad_total : 0) + a) * c.Fpad] = acc_v[a];
    }
  }
}

// Sums the per-chunk integer sums and converts them to the familiarity double:
//   fam[a][f] = P - (0.5*cw*S_hs + (1-cw)*S_v) / 255        (one rounding per operation)
// grid = (ceil(Fpad/1024), A), four views per thread; also leaves each block's maximum in blockmax[a][blockIdx.x] (no atomics:
// thousands of atomics on one 128-byte line serialise at the memory side, ~10 ns each).
__global__ void __launch_bounds__(256)
k_combine(const unsigned* __restrict__ part, const int* __restrict__ hsconst, const int* __restrict__ vconst,
          double* __restrict__ fam, unsigned long long* __restrict__ blockmax, StepState* __restrict__ st, LibCfg c, int nchunk,
          int APAD, int has_hs_sum, int has_v_sum, int n_agents) {
    __shared__ unsigned long long wmax[4];
    // four consecutive views per thread: one 16-byte load per (chunk, sum) row
    const long long f0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int a = blockIdx.y;
    const int nsum = has_hs_sum + has_v_sum;
    if (blockIdx.x == 0 && blockIdx.y == 0) reset_step_state(st, threadIdx.x, n_agents);
    unsigned long long key = 0;
    if (f0 < c.Fpad) {
        // a chunk's sum is an int32: the bit-plane path keeps the per-heading constants out of it, so it may be negative
        const long long base = acc_sum(hsconst, a);
        const long long vbase = vconst ? acc_sum(vconst, a) : 0;
        long long shs[4] = {base, base, base, base};
        long long sv[4] = {vbase, vbase, vbase, vbase};
        for (int ch = 0; ch < nchunk; ++ch) {
            const unsigned* p = part + ((long long)ch * nsum * APAD) * c.Fpad + f0;
            if (has_hs_sum) {
                const uint4 q = *reinterpret_cast<const uint4*>(p + (long long)a * c.Fpad);
                shs[0] += (int)q.x; shs[1] += (int)q.y; shs[2] += (int)q.z; shs[3] += (int)q.w;
            }
            if (has_v_sum) {
                const uint4 q = *reinterpret_cast<const uint4*>(p + (long long)((has_hs_sum ? APAD : 0) + a) * c.Fpad);
                sv[0] += (int)q.x; sv[1] += (int)q.y; sv[2] += (int)q.z; sv[3] += (int)q.w;
            }
        }
        double val[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double acc = c.whs * (double)shs[i];
            if (has_v_sum) acc = acc + c.wv * (double)sv[i];
            val[i] = (double)c.P - acc / 255.;
            if (f0 + i < c.F) {
                const unsigned long long k = ordered_key(val[i]);
                key = k > key ? k : key;
            }
        }
        double2* o = reinterpret_cast<double2*>(fam + (long long)a * c.Fpad + f0);
        o[0] = make_double2(val[0], val[1]);
        o[1] = make_double2(val[2], val[3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = wmax[0];
        for (int i = 1; i < 4; ++i) m = wmax[i] > m ? wmax[i] : m;
        blockmax[(long long)a * gridDim.x + blockIdx.x] = m;
    }
}

// ------------------------------------------------------------------ device -> host hand-over without a stream wait
// Copies n doubles to mapped host memory, then the sequence word (system scope, after everything else).
__global__ void __launch_bounds__(256)
k_publish(const double* __restrict__ src, double* __restrict__ dst, unsigned long long* __restrict__ flag, int n,
          unsigned long long seq) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ------------------------------------------------------------------ packed keys for the all-reduce(max) exchange
// The sharded step's fast path (navsim_amd/sharded.py:pack_keys): ONE all-reduce(max) of uint64 words per step,
//   keys[a], a < A                 ordered key of this rank's max_f fam[a][f]  -> the reduction is angle_familiarity
//   keys[A + 4r .. A + 4r + 3]     rank r's own slot, zero on every other rank (so the maximum is rank r's words):
//                                  ordered key of its best score; candidate count | state << 32 | 1 << 48;
//                                  its first-maximum heading + 1; that heading's first view (global index) + 1.
// Every rank can then tell whether a single (heading, view) pair lies within delta of the global maximum -- the usual
// case -- and if so which; otherwise the full records are exchanged as before.  `flip` = 1 flips the top bit of every
// word so that a SIGNED 64-bit maximum (torch's int64 all_reduce) orders them like the unsigned one.
constexpr int kKeyWordsPerRank = 4;
__global__ void __launch_bounds__(64)
k_make_keys(const double* __restrict__ rec, unsigned long long* __restrict__ keys, int A, int rank, int world, int flip) {
    const int lane = threadIdx.x;
    const unsigned long long top = flip ? 0x8000000000000000ull : 0ull;
    const int n = A + kKeyWordsPerRank * world;
    for (int i = lane; i < n; i += 64) {
        unsigned long long w = 0;
        if (i < A) w = ordered_key(rec[3 + i]);
        keys[i] = w ^ top;
    }
    // this rank's first-maximum heading over the per-heading maxima of its record
    unsigned long long k = lane < A ? ordered_key(rec[3 + lane]) : 0ull;
    unsigned long long m = k;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(m, o);
        m = other > m ? other : m;
    }
    int idx = (lane < A && k == m) ? lane : 64;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(idx, o);
        idx = other < idx ? other : idx;
    }
    __syncthreads();
    if (lane == 0) {
        if (idx > 63) idx = 0;
        unsigned long long cnt = (unsigned long long)rec[1];
        if (cnt > 0x7fffffffull) cnt = 0x7fffffffull;
        unsigned long long* slot = keys + A + kKeyWordsPerRank * rank;
        slot[0] = ordered_key(rec[0]) ^ top;
        slot[1] = (cnt | ((unsigned long long)rec[2] << 32) | (1ull << 48)) ^ top;
        slot[2] = (unsigned long long)(idx + 1) ^ top;
        slot[3] = ((unsigned long long)(long long)rec[3 + A + idx] + 1ull) ^ top;
    }
}

// ------------------------------------------------------------------ mailbox exchange (one node, host-shared memory)
// One entry per (slot, rank) of a host segment that every rank's process has mapped and registered with its GPU:
//   entry = kMboxEntry doubles: [0] sequence number, [1] check word (bits), [2 .. 2+n) the rank's packed record.
// k_post copies this rank's record into its entry (the stores go over PCIe to host memory; the other ranks' HOSTS
// poll them), payload and check word first, no fence: the reader validates the check word, as for the step record.
constexpr int kMboxEntry = 512;     // 4 KB: 3 + 4*64 record doubles fit
__global__ void __launch_bounds__(256)
k_post(const double* __restrict__ rec, double* __restrict__ entry, int n, unsigned long long seq) {
    __shared__ unsigned long long s_x;
    if (threadIdx.x == 0) s_x = 0x9E3779B97F4A7C15ull ^ (seq * 0xD1B54A32D192ED03ull);
    __syncthreads();
    unsigned long long x = 0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double v = rec[i];
        entry[2 + i] = v;
        x ^= (unsigned long long)__double_as_longlong(v) * (unsigned long long)(2 * i + 3);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x ^= __shfl_xor(x, o);
    if ((threadIdx.x & 63) == 0 && x != 0) atomicXor(&s_x, x);
    __syncthreads();
    if (threadIdx.x == 0) {
        reinterpret_cast<unsigned long long*>(entry)[1] = s_x;
        reinterpret_cast<unsigned long long*>(entry)[0] = seq;
    }
}

// ------------------------------------------------------------------ exact (sequential fp64) scoring
// fam[a][f] = the reference's value bit for bit: per-pixel terms in the reference's operation
// order, accumulated sequentially in row-major pixel order (navsim/util.pyx:44-73).
// grid = (view groups, ceil(A/4)), block = (64, 4): lane <-> view, threadIdx.y <-> heading.
// Leaves each wave's maximum in groupmax[a][g].
__global__ void __launch_bounds__(256)
k_exact_all(const uint4* __restrict__ tiles, const unsigned char* __restrict__ raw_patches,
            double* __restrict__ fam, unsigned long long* __restrict__ groupmax, StepState* __restrict__ st, LibCfg c,
            int A, int n_agents) {
    const int lane = threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0) reset_step_state(st, threadIdx.y * 64 + threadIdx.x, n_agents);
    const int a = blockIdx.y * 4 + threadIdx.y;
    if (a >= A) return;
    const long long g = blockIdx.x;
    const long long f = g * 64 + lane;
    const uint4* base = tiles + g * c.gstride + lane;
    const unsigned char* pa = raw_patches + (long long)a * c.P * 3;
    double diff = 0.0;
    for (int q = 0; q < c.Q; ++q) {
        uint4 L[kMaxHues + 1];
        for (int pl = 0; pl < c.npl; ++pl) L[pl] = base[(long long)(pl * c.Q + q) * 64];
        for (int i = 0; i < 16; ++i) {
            const int px = q * 16 + i;
            if (px >= c.P) break;
            unsigned lib[kMaxHues + 1];
            for (int pl = 0; pl < c.npl; ++pl) {
                const unsigned w = (i < 4) ? L[pl].x : (i < 8) ? L[pl].y : (i < 12) ? L[pl].z : L[pl].w;
                lib[pl] = (w >> (8 * (i & 3))) & 0xffu;
            }
            int hs, dv;
            px_ints(c, lib, pa[px * 3], pa[px * 3 + 1], pa[px * 3 + 2], hs, dv);
            diff += px_term(hs, dv, c.cw, c.wv);
        }
    }
    const double val = (double)c.P - diff;
    unsigned long long key = 0;
    if (f < c.F) {
        fam[(long long)a * c.Fpad + f] = val;
        key = ordered_key(val);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
    }
    if (lane == 0) groupmax[(long long)a * gridDim.x + g] = key;
}

// ------------------------------------------------------------------ reductions after scoring
struct StepResultDev {                   // mirrors dv_step_result (include/dejavu.h)
    int best_heading;
    unsigned flags;
    long long best_view;
    double best_fam;
    double approx_max;
    double delta;
    long long n_candidates;
    int n_headings;
    int reserved;
    double angle_fam[kMaxHeadings];
    long long angle_view[kMaxHeadings];
    double exact_fam[kMaxHeadings];
    long long exact_view[kMaxHeadings];
    unsigned long long check;                // internal (not in dv_step_result): XOR of the words k_tail wrote, see record_check
};
// Words of a record that k_tail writes for A headings: the header (7 x 8 bytes, word 6 = n_headings | seq << 32) and
// the first A entries of the four per-heading arrays.  k_tail stores them and their XOR without any fence; the
// host, polling the sequence number, accepts the record only when the XOR matches (words still in flight make it
// differ) -- a system-scope fence between payload and flag costs a round trip to host memory on every step.
__host__ __device__ inline unsigned long long record_check(const unsigned long long* w, int A) {
    unsigned long long x = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 7; ++i) x ^= w[i] * (unsigned long long)(2 * i + 3);
    for (int k = 0; k < 4; ++k)
        for (int a = 0; a < A; ++a) x ^= w[7 + k * kMaxHeadings + a] * (unsigned long long)(2 * (7 + k * kMaxHeadings + a) + 3);
    return x;
}
constexpr unsigned kResNeedsResolve = 8u;   // internal: candidates must be re-scored exactly before deciding
constexpr unsigned kResSenseError = 16u;    // the resident patches were sensed past the end of the landscape

// np.argmax over headings with the reference's first-maximum rule (NavBySceneFamiliarity.py:315),
// on exact values wherever the integer scores cannot decide.  One thread.
__device__ void decide_core(const unsigned long long* amax, const unsigned long long* aview, unsigned long long n_all,
                            bool resolved, const unsigned long long* ekey, const unsigned long long* eview,
                            StepResultDev* out, const LibCfg& c, int A, double delta, int exact_all) {
    unsigned flags = 0;
    if (resolved) flags |= 1u;
    if (exact_all) flags |= 2u;
    if (n_all > (unsigned long long)kCandCap) flags |= 4u;
    unsigned long long gkey = 0, bkey = 0;
    int best = 0;
    for (int a = 0; a < A; ++a) {
        const unsigned long long k = amax[a];
        gkey = k > gkey ? k : gkey;
        double v = key_to_double(k);
        long long view = (long long)(~aview[a]) + c.first;
        double ex = __longlong_as_double(0xfff0000000000000ll);
        long long exv = -1;
        unsigned long long dk = k;                      // key used for the decision
        if (resolved) {
            if (ekey[a]) {
                ex = key_to_double(ekey[a]);
                exv = (long long)(~eview[a]) + c.first;
                v = ex;
                view = exv;
                dk = ekey[a];
            } else {
                dk = 0;                                  // no candidate: cannot be the maximum
            }
        }
        out->angle_fam[a] = v;
        out->angle_view[a] = view;
        out->exact_fam[a] = ex;
        out->exact_view[a] = exv;
        if (dk > bkey) { bkey = dk; best = a; }          // strict '>' keeps the first maximum
    }
    out->best_heading = best;
    out->best_view = out->angle_view[best];
    out->best_fam = out->angle_fam[best];
    out->approx_max = key_to_double(gkey);
    out->delta = delta;
    out->n_candidates = (long long)n_all;
    out->n_headings = A;
    out->reserved = 0;
    out->flags = flags;
}

// decide_core for the unresolved case, spread over a 256-thread block (the serial form costs ~3 us at the end of every
// step): lane a of wave 0 fills heading a's entries, the first maximum is a wave reduction (largest key, then smallest
// index), and every thread adds its words to the record's check word.  `out` is in LDS; call with all threads.
__device__ __forceinline__ void decide_block(const unsigned long long* amax, const unsigned long long* aview,
                                             unsigned long long n_all, StepResultDev* out, unsigned long long* check,
                                             const LibCfg& c, int A, double delta, unsigned flags, int seq) {
    const int tid = threadIdx.x;
    if (tid == 0) *check = 0x9E3779B97F4A7C15ull;
    if (tid < 64) {
        const int a = tid;
        const bool valid = a < A;
        const unsigned long long k = valid ? amax[a] : 0ull;
        const long long view = valid ? (long long)(~aview[a]) + c.first : -1;
        if (valid) {
            out->angle_fam[a] = key_to_double(k);
            out->angle_view[a] = view;
            out->exact_fam[a] = __longlong_as_double(0xfff0000000000000ll);
            out->exact_view[a] = -1;
        }
        unsigned long long m = k;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long other = __shfl_xor(m, o);
            m = other > m ? other : m;
        }
        int idx = (valid && k == m) ? a : 64;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const int other = __shfl_xor(idx, o);
            idx = other < idx ? other : idx;
        }
        if (idx > 63) idx = 0;                                   // no heading has a view (cannot happen with F >= 1)
        const long long bview = __shfl(view, idx);
        if (tid == 0) {
            out->best_heading = idx;
            out->flags = flags | (n_all > (unsigned long long)kCandCap ? 4u : 0u);
            out->best_view = bview;
            out->best_fam = key_to_double(m);
            out->approx_max = key_to_double(m);
            out->delta = delta;
            out->n_candidates = (long long)n_all;
            out->n_headings = A;
            out->reserved = seq;
        }
    }
    __syncthreads();
    const unsigned long long* w = reinterpret_cast<const unsigned long long*>(out);
    unsigned long long x = 0;
    if (tid < 7) x = w[tid] * (unsigned long long)(2 * tid + 3);
    for (int i = tid; i < 4 * A; i += blockDim.x) {
        const int o = 7 + (i / A) * kMaxHeadings + (i % A);
        x ^= w[o] * (unsigned long long)(2 * o + 3);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x ^= __shfl_xor(x, o);
    if ((tid & 63) == 0 && x != 0) atomicXor(check, x);
    __syncthreads();
    if (tid == 0) out->check = *check;
    __syncthreads();
}

// Packed per-rank record for the sharded exchange (navsim_amd/sharded.py:pack_record):
//   [approx_max, n_candidates, state, angle_fam[A], angle_view[A], exact_fam[A], exact_view[A]]   (doubles)
// state: 0 integer-sum scores only, 1 candidates re-scored exactly, 2 every score exact; + 4 when the patches were
// sensed past the end of the landscape (the reference's IndexError, util.pyx:137-168).
__device__ __forceinline__ void emit_record(const StepResultDev* r, double* __restrict__ rec, int A, int tid, int nthreads) {
    if (tid == 0) {
        rec[0] = r->approx_max;
        rec[1] = (double)r->n_candidates;
        // + 4: the patches were sensed past the end of the landscape (every rank raises IndexError on seeing it)
        rec[2] = ((r->flags & 2u) ? 2.0 : ((r->flags & 1u) ? 1.0 : 0.0)) + ((r->flags & kResSenseError) ? 4.0 : 0.0);
    }
    for (int i = tid; i < A; i += nthreads) {
        rec[3 + i] = r->angle_fam[i];
        rec[3 + A + i] = (double)r->angle_view[i];
        rec[3 + 2 * A + i] = r->exact_fam[i];
        rec[3 + 3 * A + i] = (double)r->exact_view[i];
    }
}

// Everything after scoring, in one launch.  grid = ceil(F/256) blocks of 256 threads.
//  1. every block reduces the partial maxima pmax[a][0..n_partial) to amax[a] (LDS; redundant but
//     tiny, and it avoids both a grid-wide sync and same-line global atomics);
//  2. one thread per view: scene_fam[f] = min_a fam[a][f] (NavBySceneFamiliarity.py:301-303), the
//     first view attaining each heading's maximum, and the candidate list: every (a,f) whose
//     integer-sum score is within `delta` of the global maximum;
//  3. the block that finishes last decides (argmax over headings) unless more than one candidate
//     needs exact re-scoring, in which case it flags the result and the host runs k_resolve + k_decide.
// blockIdx.y = agent of a batched pass: agent g owns headings [g*A, (g+1)*A) and its own state, candidate list,
// result record and packed record.
__global__ void __launch_bounds__(256)
k_tail(const double* __restrict__ fam, const unsigned long long* __restrict__ pmax, int n_partial,
       StepState* __restrict__ st, unsigned long long* __restrict__ cand, double* __restrict__ scene,
       StepResultDev* __restrict__ out, double* __restrict__ rec, LibCfg c, int A, double delta, int want_scene,
       int exact_all, int force, int seq, const unsigned long long* __restrict__ sense_err, double delta_rel, int fenced) {
    __shared__ unsigned long long s_amax[kMaxHeadings];
    __shared__ int s_last;
    const int agent = blockIdx.y;
    const int a0 = agent * A;
    st += agent;
    cand += (long long)agent * kCandCap;
    out += agent;
    rec += (long long)agent * (3 + 4 * kMaxHeadings);
    fam += (long long)a0 * c.Fpad;
    pmax += (long long)a0 * n_partial;
    // amax[a], redundantly in every block (no grid-wide sync, no global atomics).  Many partial maxima (one per view
    // group after k_exact_all): wave w folds headings w, w+4, ...; lanes stride over them with eight loads in flight
    if (A * n_partial <= 4096) {
        // few partial maxima (k_combine's per-block ones): one round of independent loads, folded with LDS atomics
        if (threadIdx.x < kMaxHeadings) s_amax[threadIdx.x] = 0;
        __syncthreads();
        for (int i = threadIdx.x; i < A * n_partial; i += blockDim.x) atomicMax(&s_amax[i / n_partial], pmax[i]);
    } else {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int a = wave; a < A; a += 4) {
            const unsigned long long* row = pmax + (long long)a * n_partial;
            unsigned long long m = 0;
            for (int i0 = lane; i0 < n_partial; i0 += 64 * 8) {
                unsigned long long v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = row[(i0 + 64 * k < n_partial) ? i0 + 64 * k : n_partial - 1];   // clamped: a maximum does not mind repeats
#pragma unroll
                for (int k = 0; k < 8; ++k) m = v[k] > m ? v[k] : m;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const unsigned long long other = __shfl_xor(m, o);
                m = other > m ? other : m;
            }
            if (lane == 0) s_amax[a] = m;
        }
    }
    __syncthreads();

    // (views block-stride; the host launches one view per thread)
    unsigned long long gkey = 0;
    for (int a = 0; a < A; ++a) gkey = s_amax[a] > gkey ? s_amax[a] : gkey;
    const double gbest = key_to_double(gkey);
    const double thr = gbest - delta - delta_rel * fabs(gbest);
    for (long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x; f < c.F; f += (long long)gridDim.x * blockDim.x) {
        double smin = __longlong_as_double(0x7ff0000000000000ll);
        for (int a0 = 0; a0 < A; a0 += 16) {
            double v[16];                                   // all loads of a tile issued before the first use
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = fam[(long long)(a0 + k < A ? a0 + k : A - 1) * c.Fpad + f];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int a = a0 + k;
                if (a < A) {
                    smin = v[k] < smin ? v[k] : smin;
                    if (ordered_key(v[k]) == s_amax[a]) atomicMax(&st->aview[a], ~(unsigned long long)f);
                    if (!exact_all && v[k] >= thr) {
                        const unsigned long long pos = atomicAdd(&st->ncand, 1ull);
                        if (pos < (unsigned long long)kCandCap) cand[pos] = ((unsigned long long)a << 40) | (unsigned long long)f;
                    }
                }
            }
        }
        if (want_scene) scene[f] = smin;
    }

    // Arrival ticket.  Everything the last block reads from the others (aview[], ncand) was written with device-scope
    // atomics, which are performed at the device's coherence point, and every wave waits for its own to be
    // acknowledged before the workgroup's ticket is taken; the plain stores (cand[], scene[]) are only read after
    // the kernel boundary.  So no release fence here and no acquire in the last block: on this multi-XCD part they
    // are an L2 write-back / invalidate each (-1.5 us per step; tools/stress_tail.py has checked the decisions of
    // 2.5 million steps x 196 blocks against known answers).
    // `fenced` puts the release / acquire pair of the HIP memory model back (an L2 write-back and an L1 invalidate on this
    // part, ~1.5 us): the engine asks for it wherever that time does not matter (exact mode, ssd_f32, batched passes,
    // DEJAVU_FENCED=1) and the GPU suite checks that both forms decide alike.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        s_last = (atomicAdd(&st->done, 1u) == gridDim.x - 1) ? 1 : 0;
        if (fenced && s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!s_last) return;

    // Last block: decide, build the record in LDS, then copy it out with wide coalesced stores
    // (the record lives in mapped host memory: a few PCIe writes instead of one per field).
    __shared__ StepResultDev s_res;
    __shared__ unsigned long long s_aview[kMaxHeadings];
    __shared__ unsigned long long s_ncand;
    __shared__ int s_serr;
    if (threadIdx.x < A) {
        __hip_atomic_store(&st->amax[threadIdx.x], s_amax[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for k_decide on the resolve path
        s_aview[threadIdx.x] = __hip_atomic_load(&st->aview[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // the other two things the decision needs, fetched by other waves in the same round trip
    if (threadIdx.x == 64) s_ncand = __hip_atomic_load(&st->ncand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x == 128) s_serr = sense_err ? (int)((*sense_err >> agent) & 1ull) : 0;   // bit = agent of the pass
    __syncthreads();
    {
        const unsigned long long n_all = s_ncand;
        const bool overflow = n_all > (unsigned long long)kCandCap;
        const bool needs = !exact_all && !overflow && (n_all >= 2 || (force && n_all >= 1));
        // the integer-score decision is always filled in (the sharded exchange wants the per-heading maxima
        // even when local near-ties still have to be re-scored); NEEDS_RESOLVE tells the host it is provisional.
        // Spread over the block like k_fold's (decide_block: the one-thread form was ~3 us at the end of every such step); the
        // record's `reserved` word carries seq: the host may poll it instead of waiting for the stream.
        __shared__ unsigned long long s_check;
        decide_block(s_amax, s_aview, n_all, &s_res, &s_check, c, A, delta,
                     (exact_all ? 2u : 0u) | (needs ? kResNeedsResolve : 0u) | (s_serr ? kResSenseError : 0u), seq);      // (patches from k_sense that ran off the landscape)
        if (threadIdx.x == 0)
            __hip_atomic_store(&st->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // k_finish, which may run the next step, expects it clear
    }
    emit_record(&s_res, rec, A, threadIdx.x, blockDim.x);
    // The record goes to mapped host memory with wide coalesced stores and no fence: header (7 words), the first A
    // entries of each of the four per-heading arrays, and the check word (see record_check).
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&s_res);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(out);
    if (threadIdx.x < 7) dst[threadIdx.x] = src[threadIdx.x];
    if (threadIdx.x == 7) out->check = s_res.check;
    for (int i = threadIdx.x; i < 4 * A; i += blockDim.x) {
        const int o = 7 + (i / A) * kMaxHeadings + (i % A);
        dst[o] = src[o];
    }
}

// ------------------------------------------------------------------ k_finish: combine + tail in ONE launch
// The integer path's steps end here (k_combine + k_tail remain for dv_score, the exact mode and ssd_f32).  Each block
// owns 256 views: it turns the scoring kernel's partial sums into scores (k_combine's arithmetic) and keeps them in
// registers, so fam[] is neither written nor read back.  What k_tail needs the GLOBAL maxima for is rearranged so
// that every block can finish on its own and only the last one to arrive combines:
//   * per heading, the block leaves its maximum and the first view attaining it (bsum);
//   * candidates for exact re-scoring are all (a,f) within delta of the global maximum.  A block does not know
//     that maximum, but it is at least the block's own: every entry within delta of the BLOCK's best that is not
//     already one of its per-heading representatives goes to a shared list (normally none does);
//   * the last block folds the summaries into amax[a] / first view, derives the true threshold, and builds the
//     candidate list from the representatives and the shared list, dropping what falls short -- the same set
//     k_tail lists.  It then decides and writes the record as k_tail does, and clears the counters for the next step.
// Cross-block data travels through device-scope atomic stores/loads (no fences, see k_tail).
constexpr int kTmpCap = 4096;       // entries of the shared extra-candidate list per agent

// The part of k_finish after every block has left its summary: fold the summaries into per-heading maxima and first
// views, derive the threshold, build the candidate list from the representatives and the shared list, decide, write the
// result record.  Run by the last block to arrive (small libraries) or by k_fold, a kernel of its own with 1024 threads
// behind k_finish (large libraries: the kernel boundary replaces the arrival ticket, and four times the threads walk the
// summaries).  Generic in blockDim.x.
template <int kBatch = 16>
__device__ __forceinline__ void fold_and_decide(const unsigned long long* __restrict__ bsum, const unsigned long long* __restrict__ ctmp,
                                                unsigned long long* __restrict__ cand, StepState* __restrict__ st,
                                                StepResultDev* __restrict__ out, double* __restrict__ rec, const LibCfg& c, int A,
                                                double delta, int force, int seq, const unsigned long long* __restrict__ sense_err,
                                                int agent, int nb) {
    const int tid = threadIdx.x;
    // ---- last block: fold the summaries, list the candidates, decide
    __shared__ unsigned long long s_amax[kMaxHeadings];
    __shared__ unsigned long long s_aview[kMaxHeadings];
    __shared__ unsigned s_ncount;
    __shared__ unsigned s_ntmp;
    __shared__ int s_serr;
    __shared__ StepResultDev s_res;
    const int G = blockDim.x / A;                       // thread groups; thread (a, r) walks blocks r, r+G, ...
    const int a = tid % A, r = tid / A;
    const bool active = r < G;
    // kBatch: summaries requested before the first is used (16; 8 in the 1024-thread k_fold, whose 128 registers per lane 16 did not fit)
    // Up to kBatch blocks per thread (50 000 views x 16 headings: 13): maxima AND first views are fetched in one
    // round trip and stay in registers for the second pass.  Longer lists are walked twice.
    const bool single = nb <= G * kBatch;
    unsigned long long k8[kBatch], v8[kBatch];
#pragma unroll
    for (int j = 0; j < kBatch; ++j) { k8[j] = 0; v8[j] = ~0ull; }
    if (active && single) {                             // (requested in front of the barrier: in the same round trip as ntmp and the error bit)
#pragma unroll
        for (int j = 0; j < kBatch; ++j) {
            const int b = (r + j * G < nb) ? r + j * G : nb - 1;                   // clamped: no conditional loads
            k8[j] = __hip_atomic_load(&bsum[((long long)b * 2 + 0) * A + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v8[j] = __hip_atomic_load(&bsum[((long long)b * 2 + 1) * A + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid < kMaxHeadings) { s_amax[tid] = 0; s_aview[tid] = ~0ull; }
    if (tid == 0) s_ncount = 0;
    if (tid == 64) s_ntmp = __hip_atomic_load(&st->ntmp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid == 128) s_serr = sense_err ? (int)((*sense_err >> agent) & 1ull) : 0;          // bit = agent of the pass
    __syncthreads();
    if (active) {
        unsigned long long lk = 0;
        if (single) {
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                if (r + j * G >= nb) k8[j] = 0;                                     // a clamped repeat: not an entry
                lk = k8[j] > lk ? k8[j] : lk;
            }
        } else {
            for (int b0 = r; b0 < nb; b0 += G * kBatch) {
                unsigned long long t8[kBatch];
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    const int b = (b0 + j * G < nb) ? b0 + j * G : nb - 1;
                    t8[j] = __hip_atomic_load(&bsum[((long long)b * 2 + 0) * A + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < kBatch; ++j) lk = t8[j] > lk ? t8[j] : lk;
            }
        }
        atomicMax(&s_amax[a], lk);
    }
    __syncthreads();
    unsigned long long gkey = 0;
    for (int k = 0; k < A; ++k) gkey = s_amax[k] > gkey ? s_amax[k] : gkey;
    const unsigned long long thr_key = ordered_key(key_to_double(gkey) - delta);
    if (active) {
        // second pass: blocks holding this heading's maximum give its first view; blocks whose maximum reaches the
        // threshold (normally one in all) give a candidate
        const unsigned long long amax_a = s_amax[a];
        if (single) {
#pragma unroll
            for (int j = 0; j < kBatch; ++j) {
                const unsigned long long key = k8[j];
                if (key != 0 && (key == amax_a || key >= thr_key)) {
                    if (key == amax_a) atomicMin(&s_aview[a], v8[j]);
                    if (key >= thr_key) {
                        const unsigned pos = atomicAdd(&s_ncount, 1u);
                        if (pos < (unsigned)kCandCap) cand[pos] = ((unsigned long long)a << 40) | v8[j];
                    }
                }
            }
        } else {
            for (int b0 = r; b0 < nb; b0 += G * kBatch) {              // the maxima again, a batch per round trip
                unsigned long long t8[kBatch];
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    const int b = (b0 + j * G < nb) ? b0 + j * G : nb - 1;
                    t8[j] = __hip_atomic_load(&bsum[((long long)b * 2 + 0) * A + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
#pragma unroll
                for (int j = 0; j < kBatch; ++j) {
                    const int b = b0 + j * G;
                    const unsigned long long key = t8[j];
                    if (b < nb && key != 0 && (key == amax_a || key >= thr_key)) {   // few blocks qualify: their views one by one
                        const unsigned long long view = __hip_atomic_load(&bsum[((long long)b * 2 + 1) * A + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (key == amax_a) atomicMin(&s_aview[a], view);
                        if (key >= thr_key) {
                            const unsigned pos = atomicAdd(&s_ncount, 1u);
                            if (pos < (unsigned)kCandCap) cand[pos] = ((unsigned long long)a << 40) | view;
                        }
                    }
                }
            }
        }
    }
    const unsigned n_tmp = s_ntmp < (unsigned)kTmpCap ? s_ntmp : (unsigned)kTmpCap;
    for (unsigned i = tid; i < n_tmp; i += blockDim.x) {
        const unsigned long long e = __hip_atomic_load(&ctmp[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long k = __hip_atomic_load(&ctmp[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (k >= thr_key) {
            const unsigned pos = atomicAdd(&s_ncount, 1u);
            if (pos < (unsigned)kCandCap) cand[pos] = e;
        }
    }
    __syncthreads();
    if (tid < A) {
        s_aview[tid] = ~s_aview[tid];                                          // decide_core / k_decide expect ~f (0 = none)
        __hip_atomic_store(&st->amax[tid], s_amax[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->aview[tid], s_aview[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    // the shared list overflowing means candidates were lost: report more than the resolver can take
    const unsigned long long n_all = (s_ntmp > (unsigned)kTmpCap) ? (unsigned long long)kCandCap + 1 : (unsigned long long)s_ncount;
    const bool needs = n_all <= (unsigned long long)kCandCap && (n_all >= 2 || (force && n_all >= 1));
    if (tid == 0) {                                                            // counters clear for the next step; every
        // location other blocks touch atomically is also reset atomically (agent scope)
        __hip_atomic_store(&st->ncand, n_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->ntmp, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __shared__ unsigned long long s_check;
    decide_block(s_amax, s_aview, n_all, &s_res, &s_check, c, A, delta, (needs ? kResNeedsResolve : 0u) | (s_serr ? kResSenseError : 0u), seq);
    emit_record(&s_res, rec, A, tid, blockDim.x);
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&s_res);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(out);
    if (tid < 7) dst[tid] = src[tid];
    if (tid == 7) out->check = s_res.check;
    for (int i = tid; i < 4 * A; i += blockDim.x) {
        const int o = 7 + (i / A) * kMaxHeadings + (i % A);
        dst[o] = src[o];
    }
}

template <int NT>                   // headings per agent <= 16 * NT, scores held in registers
__global__ void __launch_bounds__(256)
k_finish(const unsigned* __restrict__ part, const int* __restrict__ hsconst, const int* __restrict__ vconst, int nchunk,
         int APAD, int has_hs_sum, int has_v_sum, StepState* __restrict__ st, unsigned long long* __restrict__ bsum,
         unsigned long long* __restrict__ ctmp, unsigned long long* __restrict__ cand, double* __restrict__ scene,
         StepResultDev* __restrict__ out, double* __restrict__ rec, LibCfg c, int A, double delta, int want_scene, int force,
         int seq, const unsigned long long* __restrict__ sense_err, int fenced, int vb, int separate_fold) {
    __shared__ unsigned long long s_bmax[kMaxHeadings];
    __shared__ unsigned long long s_bview[kMaxHeadings];
    __shared__ unsigned long long s_keys[16 * 16 * 17];          // 34 KB: key transposes of phase 2
    __shared__ int s_last;
    const int agent = blockIdx.y;
    const int a_base = agent * A;
    const int nb = gridDim.x;
    const int tid = threadIdx.x;
    st += agent;
    cand += (long long)agent * kCandCap;
    ctmp += (long long)agent * kTmpCap * 2;
    bsum += (long long)agent * nb * 2 * A;
    out += agent;
    rec += (long long)agent * (3 + 4 * kMaxHeadings);

    // ---- scores of this thread's views, all headings of the agent.  A block owns `vb` consecutive sets of 256 views
    // (vb > 1 on large libraries: the last block's fold walks one summary per block, so fewer, larger blocks).
    const int nsum = has_hs_sum + has_v_sum;
    // Sixteen headings at a time (NT rounds), so that a thread never holds more than 16 scores: each round loads its
    // integer sums (per chunk, all 16 headings' loads issued together), turns them into scores, finds the block's
    // maximum and first view per heading through an LDS transpose, and lists candidates against the round's OWN best --
    // a threshold that can only be lower than the block's or the global one, so the list stays a superset of what the
    // last block keeps after it has derived the true threshold.
    __shared__ unsigned long long s_blkmax[kMaxHeadings];        // over the view sets done so far (s_bmax / s_bview: current set)
    __shared__ unsigned long long s_blkview[kMaxHeadings];
    if (tid < kMaxHeadings) { s_blkmax[tid] = 0; s_blkview[tid] = ~0ull; }
#pragma unroll 1
  for (int vs = 0; vs < vb; ++vs) {
    if (tid < kMaxHeadings) { s_bmax[tid] = 0; s_bview[tid] = ~0ull; }
    const long long f = ((long long)blockIdx.x * vb + vs) * blockDim.x + tid;
    const bool inb = f < c.F;
    const long long fl = inb ? f : c.F - 1;
    double smin = __longlong_as_double(0x7ff0000000000000ll);
#pragma unroll 1
    for (int t = 0; t < NT; ++t) {
        if (t * 16 >= A) break;
        const int At = (A - t * 16) < 16 ? (A - t * 16) : 16;                   // headings of this round
        unsigned shs_u[16], sv_u[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) { shs_u[k] = 0; sv_u[k] = 0; }
        for (int ch = 0; ch < nchunk; ++ch) {
            const unsigned* p = part + ((long long)ch * nsum * APAD) * c.Fpad + fl;
            unsigned th[16], tv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int a = a_base + t * 16 + (k < At ? k : At - 1);          // clamped: no conditional loads
                th[k] = has_hs_sum ? p[(long long)a * c.Fpad] : 0u;
                tv[k] = has_v_sum ? p[(long long)((has_hs_sum ? APAD : 0) + a) * c.Fpad] : 0u;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) { shs_u[k] += th[k]; sv_u[k] += tv[k]; }
        }
        double val[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            val[k] = 0.0;
            if (k < At) {
                // the chunk sums are int32 and may wrap on the way (bit-plane path: negative chunks); their total fits
                const long long shs = (long long)acc_sum(hsconst, a_base + t * 16 + k) + (long long)(int)shs_u[k];
                const long long sv = (long long)(vconst ? acc_sum(vconst, a_base + t * 16 + k) : 0) + (long long)(int)sv_u[k];
                double acc = c.whs * (double)shs;
                if (has_v_sum) acc = acc + c.wv * (double)sv;
                val[k] = (double)c.P - acc / 255.;
            }
        }
        // keys through LDS transposed: thread (k = tid/16, j = tid%16) folds the 16 keys of heading k from views
        // j*16..j*16+15 (rows padded to 17 against bank conflicts), then the 16 partial results per heading meet in LDS atomics
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
            s_keys[(kk * 16 + (tid >> 4)) * 17 + (tid & 15)] = inb ? ordered_key(val[kk]) : 0ull;
        __syncthreads();
        {
            const int k = tid >> 4, j = tid & 15;
            unsigned long long m = 0;
            int mi = 0;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const unsigned long long x = s_keys[(k * 16 + j) * 17 + i];
                if (x > m) { m = x; mi = i; }
            }
            if (k < At && m != 0) atomicMax(&s_bmax[t * 16 + k], m);
            __syncthreads();
            if (k < At && m != 0 && m == s_bmax[t * 16 + k])
                atomicMin(&s_bview[t * 16 + k], (unsigned long long)(((long long)blockIdx.x * vb + vs) * blockDim.x + j * 16 + mi));
        }
        __syncthreads();
        unsigned long long rbest = 0;
        for (int k = 0; k < At; ++k) rbest = s_bmax[t * 16 + k] > rbest ? s_bmax[t * 16 + k] : rbest;
        const double thr_b = key_to_double(rbest) - delta;
        if (inb) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k < At) {
                    const int a = t * 16 + k;
                    smin = val[k] < smin ? val[k] : smin;
                    if (val[k] >= thr_b && !(ordered_key(val[k]) == s_bmax[a] && (unsigned long long)f == s_bview[a])) {
                        const unsigned pos = __hip_atomic_fetch_add(&st->ntmp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (pos < (unsigned)kTmpCap) {
                            __hip_atomic_store(&ctmp[2 * pos], ((unsigned long long)a << 40) | (unsigned long long)f, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(&ctmp[2 * pos + 1], ordered_key(val[k]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
            }
        }
    }
    if (inb && want_scene) scene[f] = smin;
    __syncthreads();
    if (tid < A && s_bmax[tid] != 0) {
        // This set's representative of heading tid against the block's so far: larger key, then smaller view, wins.  The
        // loser was kept out of the shared list as a representative, so it is listed now if it lies within delta of the
        // winner (anything within delta of the GLOBAL maximum does: the winner is no larger than that maximum).
        const unsigned long long ck = s_bmax[tid], cv = s_bview[tid], bk = s_blkmax[tid], bv = s_blkview[tid];
        if (bk == 0) {
            s_blkmax[tid] = ck; s_blkview[tid] = cv;
        } else {
            const bool cur_wins = ck > bk || (ck == bk && cv < bv);
            const unsigned long long wk = cur_wins ? ck : bk, lk = cur_wins ? bk : ck, lv = cur_wins ? bv : cv;
            if (key_to_double(lk) >= key_to_double(wk) - delta) {
                const unsigned pos = __hip_atomic_fetch_add(&st->ntmp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (pos < (unsigned)kTmpCap) {
                    __hip_atomic_store(&ctmp[2 * pos], ((unsigned long long)tid << 40) | lv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&ctmp[2 * pos + 1], lk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (cur_wins) { s_blkmax[tid] = ck; s_blkview[tid] = cv; }
        }
    }
    __syncthreads();
  }
    if (tid < A) {
        __hip_atomic_store(&bsum[((long long)blockIdx.x * 2 + 0) * A + tid], s_blkmax[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&bsum[((long long)blockIdx.x * 2 + 1) * A + tid], s_blkview[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

    if (separate_fold) return;          // k_fold, launched behind this kernel, does the rest: no ticket at all

    // ---- arrival ticket (see k_tail)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        s_last = (atomicAdd(&st->done, 1u) == (unsigned)(nb - 1)) ? 1 : 0;
        if (fenced && s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (!s_last) return;

    fold_and_decide(bsum, ctmp, cand, st, out, rec, c, A, delta, force, seq, sense_err, agent, nb);
}

// fold_and_decide as its own launch: grid (1, agents), NT threads (launch_fold picks 256 / 512 / 1024 by the length of the list).
// One instantiation per thread count, so that the 256-thread form of every single-agent step is not held to the 128 registers
// per lane of a 1024-thread workgroup (it kept six of them in scratch: tools/kernel_resources.py).
template <int NT>
__global__ void __launch_bounds__(NT)
k_fold(unsigned long long* __restrict__ bsum, unsigned long long* __restrict__ ctmp, unsigned long long* __restrict__ cand,
       StepState* __restrict__ st, StepResultDev* __restrict__ out, double* __restrict__ rec, LibCfg c, int A, double delta, int force,
       int seq, const unsigned long long* __restrict__ sense_err, int nb) {
    const int agent = blockIdx.y;
    fold_and_decide<(NT > 512 ? 8 : 16)>(bsum + (long long)agent * nb * 2 * A, ctmp + (long long)agent * kTmpCap * 2, cand + (long long)agent * kCandCap,
                    st + agent, out + agent, rec + (long long)agent * (3 + 4 * kMaxHeadings), c, A, delta, force, seq, sense_err, agent, nb);
}

// First level of a two-level fold for long summary lists (grid (slices, agents), 256 threads): a block reduces the
// summaries [slice * per, slice * per + per) of its agent to ONE summary of the same form -- per heading the maximum and
// the smallest view attaining it -- in bsum2, and lists every other representative within delta of the slice's own
// best (the superset rule of k_finish, one level up) in the shared list.  k_fold on bsum2 then walks `slices` entries
// instead of thousands with one workgroup.
constexpr int kFoldSlices = 32;
__global__ void __launch_bounds__(256)
k_fold_reduce(const unsigned long long* __restrict__ bsum, unsigned long long* __restrict__ bsum2, unsigned long long* __restrict__ ctmp,
              StepState* __restrict__ st, int A, double delta, int nb, int per) {
    const int agent = blockIdx.y, slice = blockIdx.x, tid = threadIdx.x;
    const unsigned long long* base = bsum + (long long)agent * nb * 2 * A;
    const int b0 = slice * per, b1 = (b0 + per < nb) ? b0 + per : nb;
    __shared__ unsigned long long s_max[kMaxHeadings], s_view[kMaxHeadings];
    if (tid < kMaxHeadings) { s_max[tid] = 0; s_view[tid] = ~0ull; }
    __syncthreads();
    const int G = blockDim.x / A, a = tid % A, r = tid / A;
    const bool active = r < G;
    unsigned long long lk = 0;
    if (active)
        for (int b = b0 + r; b < b1; b += G) {
            const unsigned long long k = base[((long long)b * 2 + 0) * A + a];
            lk = k > lk ? k : lk;
        }
    if (active && lk) atomicMax(&s_max[a], lk);
    __syncthreads();
    unsigned long long gkey = 0;
    for (int k = 0; k < A; ++k) gkey = s_max[k] > gkey ? s_max[k] : gkey;
    const unsigned long long thr = gkey ? ordered_key(key_to_double(gkey) - delta) : ~0ull;
    const unsigned long long amax_a = active ? s_max[a] : 0;
    if (active && amax_a)
        for (int b = b0 + r; b < b1; b += G)
            if (base[((long long)b * 2 + 0) * A + a] == amax_a) atomicMin(&s_view[a], base[((long long)b * 2 + 1) * A + a]);
    __syncthreads();
    if (active && amax_a >= thr)                              // (a heading whose best falls short of the threshold lists nothing)
        for (int b = b0 + r; b < b1; b += G) {
            const unsigned long long k = base[((long long)b * 2 + 0) * A + a];
            if (k != 0 && k >= thr) {
                const unsigned long long view = base[((long long)b * 2 + 1) * A + a];
                if (!(k == amax_a && view == s_view[a])) {
                    const unsigned pos = __hip_atomic_fetch_add(&st[agent].ntmp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (pos < (unsigned)kTmpCap) {
                        unsigned long long* ct = ctmp + (long long)agent * kTmpCap * 2;
                        ct[2 * pos] = ((unsigned long long)a << 40) | view;
                        ct[2 * pos + 1] = k;
                    }
                }
            }
        }
    if (tid < A) {
        unsigned long long* dst = bsum2 + ((long long)agent * gridDim.x + slice) * 2 * A;
        dst[tid] = s_max[tid];
        dst[A + tid] = s_max[tid] ? s_view[tid] : ~0ull;
    }
}

// One single-wave block per candidate (a,f): the reference's exact value.  Lanes compute the per-pixel
// terms of 1024 pixels at a time into LDS; the sequential double accumulation over them is
// then done identically by every lane (broadcast reads).
__global__ void __launch_bounds__(64)
k_resolve(const uint4* __restrict__ tiles, const unsigned char* __restrict__ raw_patches,
          const StepState* __restrict__ st, const unsigned long long* __restrict__ cand,
          double* __restrict__ cand_exact, LibCfg c) {
    __shared__ double terms[1024];
    const unsigned long long n_all = st->ncand;
    if (n_all > (unsigned long long)kCandCap) return;       // overflow: host redoes the step in exact mode
    const int n = (int)n_all;
    const int lane = threadIdx.x;
    for (int ci = blockIdx.x; ci < n; ci += gridDim.x) {
        const unsigned long long cf = cand[ci];
        const int a = (int)(cf >> 40);
        const long long f = (long long)(cf & 0xffffffffffull);
        const uint4* base = tiles + (f >> 6) * c.gstride + (f & 63);
        const unsigned char* pa = raw_patches + (long long)a * c.P * 3;
        double diff = 0.0;
        for (int qb = 0; qb < c.Q; qb += 64) {
            const int q = qb + lane;
            if (q < c.Q) {
                uint4 L[kMaxHues + 1];
                for (int pl = 0; pl < c.npl; ++pl) L[pl] = base[(long long)(pl * c.Q + q) * 64];
                for (int i = 0; i < 16; ++i) {
                    const int px = q * 16 + i;
                    double t = 0.0;
                    if (px < c.P) {
                        unsigned lib[kMaxHues + 1];
                        for (int pl = 0; pl < c.npl; ++pl) {
                            const unsigned w = (i < 4) ? L[pl].x : (i < 8) ? L[pl].y : (i < 12) ? L[pl].z : L[pl].w;
                            lib[pl] = (w >> (8 * (i & 3))) & 0xffu;
                        }
                        int hs, dv;
                        px_ints(c, lib, pa[px * 3], pa[px * 3 + 1], pa[px * 3 + 2], hs, dv);
                        t = px_term(hs, dv, c.cw, c.wv);
                    }
                    terms[lane * 16 + i] = t;
                }
            }
            __syncthreads();
            int npx = c.P - qb * 16;
            npx = npx > 1024 ? 1024 : npx;
            // same sequential order as the reference; the LDS reads are batched so that only the adds are serial
            int i = 0;
            for (; i + 16 <= npx; i += 16) {
                double t[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) t[k] = terms[i + k];
#pragma unroll
                for (int k = 0; k < 16; ++k) diff += t[k];
            }
            for (; i < npx; ++i) diff += terms[i];
            __syncthreads();
        }
        if (lane == 0) cand_exact[ci] = (double)c.P - diff;
    }
}

// Resolve path only: single wave; folds the exact candidate values into per-heading maxima, then decides.
__global__ void k_decide(const StepState* __restrict__ st, const unsigned long long* __restrict__ cand,
                         const double* __restrict__ cand_exact, StepResultDev* __restrict__ out,
                         double* __restrict__ rec, LibCfg c, int A, double delta,
                         const unsigned long long* __restrict__ sense_err, int agent, int seq) {
    __shared__ unsigned long long ekey[kMaxHeadings];
    __shared__ unsigned long long eview[kMaxHeadings];
    __shared__ StepResultDev s_res;
    const int lane = threadIdx.x;
    if (lane < kMaxHeadings) { ekey[lane] = 0; eview[lane] = 0; }
    __syncthreads();
    const unsigned long long n_all = st->ncand;
    const int n = n_all > (unsigned long long)kCandCap ? 0 : (int)n_all;
    for (int i = lane; i < n; i += blockDim.x) atomicMax(&ekey[cand[i] >> 40], ordered_key(cand_exact[i]));
    __syncthreads();
    for (int i = lane; i < n; i += blockDim.x) {
        const int a = (int)(cand[i] >> 40);
        if (ordered_key(cand_exact[i]) == ekey[a]) atomicMax(&eview[a], ~(cand[i] & 0xffffffffffull));
    }
    __syncthreads();
    if (lane == 0) {
        decide_core(st->amax, st->aview, n_all, n > 0, ekey, eview, &s_res, c, A, delta, 0);
        if (sense_err && ((*sense_err >> agent) & 1ull)) s_res.flags |= kResSenseError;
        if (seq >= 0) {                                  // a step that ENDS here (ssd_f32 on the matrix cores): the host polls this record
            s_res.reserved = seq;
            s_res.check = record_check(reinterpret_cast<const unsigned long long*>(&s_res), A);
        }
    }
    __syncthreads();
    emit_record(&s_res, rec, A, lane, blockDim.x);
    const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&s_res);
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(out);
    if (lane < 7) dst[lane] = src[lane];
    if (lane == 7 && seq >= 0) out->check = s_res.check;
    for (int i = lane; i < 4 * A; i += blockDim.x) {
        const int o = 7 + (i / A) * kMaxHeadings + (i % A);
        dst[o] = src[o];
    }
}

// ------------------------------------------------------------------ ssd_f32 metric
// Sum of squared differences of single-channel float32 views: the reference's only definition of "SSD" is
// navsim/util.pyx:171-184 (`ssds`, dead code there): sum over i,j of (a[i,j]-b[i,j])**2, double, row-major, sequential.
// Layout: ftiles[g][q][lane] = 4 consecutive pixels (float4) of view g*64+lane, pixels zero padded to a multiple of 4
// (a padded pixel is 0 in both operands and adds nothing); fprep[q][j][APAD] = patch pixel 4q+j of each heading.
// Accumulation: fp32 fma over runs of 8 pixels, each run then added into a double (keeps the result within 1e-6 relative of
// the reference's all-double sum, typically 1e-7; near-ties are re-scored exactly by k_resolve_f32).  Scores are stored NEGATED
// (fam = -ssd) so that the max-based reductions of k_combine_f32 / k_tail apply unchanged: most familiar = least SSD.
__global__ void k_retile_f32(const float* __restrict__ raw, float4* __restrict__ ftiles, LibCfg c) {
    const long long total = (c.Fpad / 64) * (long long)c.Q * 64;       // c.Q = ceil(P/4) for this metric
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const long long r = t >> 6;
    const int q = (int)(r % c.Q);
    const long long f = (r / c.Q) * 64 + lane;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (f < c.F)
        for (int i = 0; i < 4; ++i) { const int px = q * 4 + i; if (px < c.P) v[i] = raw[f * (long long)c.P + px]; }
    ftiles[(r / c.Q) * c.gstride + (long long)q * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
}

// Synthetic float32 library straight into ftiles: view f, pixel p <- the top 24 bits of splitmix64((first+f)*P + p + seed*GOLDEN)
// as a float in [0, 1) (exact in fp32) -- navsim_amd/synth.py:synth_views_f32 makes the same values.
__global__ void k_generate_tiles_f32(float4* __restrict__ ftiles, LibCfg c, unsigned long long seed) {
    const long long total = (c.Fpad / 64) * (long long)c.Q * 64;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const long long r = t >> 6;
    const int q = (int)(r % c.Q);
    const long long f = (r / c.Q) * 64 + lane;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (f < c.F) {
        const unsigned long long base = (unsigned long long)(c.first + f) * (unsigned long long)c.P + seed * 0x9E3779B97F4A7C15ull;
        for (int i = 0; i < 4; ++i) {
            const int px = q * 4 + i;
            if (px < c.P) v[i] = (float)(splitmix64(base + (unsigned long long)px) >> 40) * (1.0f / 16777216.0f);
        }
    }
    ftiles[(r / c.Q) * c.gstride + (long long)q * 64 + lane] = make_float4(v[0], v[1], v[2], v[3]);
}

// fprep[q][j][a] = patch pixel 4q + j of heading a: wave-uniform, read with scalar loads into SGPRs.
__global__ void k_prep_f32(const float* __restrict__ raw, float* __restrict__ fprep, LibCfg c, int A, int APAD) {
    const long long total = (long long)c.Q * 4 * APAD;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int a = (int)(t % APAD);
    const int px = (int)(t / APAD);
    fprep[t] = (a < A && px < c.P) ? raw[(long long)a * c.P + px] : 0.f;
}

// Scoring kernel of the ssd_f32 metric.  Item = (pixel chunk, view group of 64), scored by a workgroup of NW waves that
// take the item's 16-pixel blocks round-robin; lane <-> view, APAD headings per pass (a_off selects the slice of the
// apad_total resident ones: 32 or 64 headings are further passes over the library).
//   * arithmetic: d = l - p, run = fma(d, d, run) in fp32 over one 16-pixel block, then added into a double per heading
//     (~1e-7 relative; near-ties are re-scored exactly by k_resolve_f32).  Two VALU operations per pixel and heading are
//     the floor of this form (the expansion l^2 - 2 l p + p^2 halves them but cancels: a near match's small SSD would lose its
//     1e-6).  hipcc's SLP pass pairs them into v_pk_add_f32 / v_pk_fma_f32 (256 packed instructions per 16-pixel block and 16
//     headings), which is what runs fastest: 50 000 views x 64x64 x 16 headings 195 us, 245 us built with -fno-slp-vectorize
//     (round 3; hand-packed source measured slower than either in round 1).  At 16 headings the kernel is bound by that
//     instruction stream, not by HBM (8 lane-operations per library byte): the next block's tiles are loaded before the current
//     one is scored so that the loads hide behind it;
//   * the NW waves add their doubles up in LDS in a fixed order (wave 0 + 1 + 2 + 3: reproducible) and share the
//     stores, so a quarter of the partial sums cross HBM.
template <int APAD, int NW, bool PF>
__global__ void __launch_bounds__(64 * NW) __attribute__((amdgpu_num_sgpr(96))) __attribute__((amdgpu_waves_per_eu(APAD == 16 ? 4 : 6)))
k_ssd_tiles(const float4* __restrict__ ftiles, const float* __restrict__ fprep, double* __restrict__ part, LibCfg c, int nchunk,
            int apad_total, int a_off) {
    extern __shared__ double red_f[];            // [NW][APAD][64] when NW > 1
    const int lane = threadIdx.x & 63;
    const int wave = NW == 1 ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int Q = c.Q;
    const long long G = c.Fpad / 64;
    const long long n_items = G * nchunk;
    const int Q4 = (Q + 3) / 4;                  // 16-pixel blocks
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ch = (int)(item / G);
        const long long g = item - (long long)ch * G;
        const int b0 = (int)(((long long)ch * Q4) / nchunk), b1 = (int)(((long long)(ch + 1) * Q4) / nchunk);
        const float4* base = ftiles + g * c.gstride + lane;
        double acc[APAD];
#pragma unroll
        for (int a = 0; a < APAD; ++a) acc[a] = 0.0;
        auto load_block = [&](int blk, v4u_t (&dst)[4]) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int q = (blk * 4 + s < Q) ? blk * 4 + s : Q - 1;
                dst[s] = __builtin_nontemporal_load(reinterpret_cast<const v4u_t*>(&base[(long long)q * 64]));
            }
        };
        v4u_t cur[4], nxt[4];
        if (PF && b0 + wave < b1) load_block(b0 + wave, cur);
        for (int blk = b0 + wave; blk < b1; blk += NW) {
            const int qb = blk * 4;
            if (PF) load_block(blk + NW < b1 ? blk + NW : blk, nxt);  // the next block's tiles, in flight while this one is scored
            else load_block(blk, cur);
            // two fp32 runs of 8 pixels each per block (chunks 0, 1 and 2, 3), each added into the double on its own: a run's
            // rounding error is bounded by 8 fp32 roundings of its partial sums (< 5e-7 relative, typically 1e-7), which keeps a
            // score within the north star's 1e-6 of the reference's all-double sum
            float run[2][APAD];
#pragma unroll
            for (int a = 0; a < APAD; ++a) { run[0][a] = 0.f; run[1][a] = 0.f; }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (qb + s < Q) {
                    const float* pp = fprep + ((long long)(qb + s) * 4) * apad_total + a_off;          // wave-uniform -> s_load
                    const float lw[4] = {__uint_as_float(cur[s].x), __uint_as_float(cur[s].y), __uint_as_float(cur[s].z), __uint_as_float(cur[s].w)};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#pragma unroll
                        for (int a = 0; a < APAD; ++a) {
                            const float d = lw[j] - pp[j * apad_total + a];
                            run[s >> 1][a] = __builtin_fmaf(d, d, run[s >> 1][a]);
                        }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < APAD; ++a) { acc[a] += (double)run[0][a]; acc[a] += (double)run[1][a]; }
            if (PF) {
#pragma unroll
                for (int s = 0; s < 4; ++s) cur[s] = nxt[s];
            }
        }
        double* dst = part + ((long long)ch * apad_total + a_off) * c.Fpad + g * 64 + lane;
        if (NW == 1) {
#pragma unroll
            for (int a = 0; a < APAD; ++a) dst[(long long)a * c.Fpad] = acc[a];
            continue;
        }
#pragma unroll
        for (int a = 0; a < APAD; ++a) red_f[(wave * APAD + a) * 64 + lane] = acc[a];
        __syncthreads();
        for (int a = wave; a < APAD; a += NW) {
            double t = red_f[a * 64 + lane];
#pragma unroll
            for (int w2 = 1; w2 < NW; ++w2) t += red_f[(w2 * APAD + a) * 64 + lane];
            dst[(long long)a * c.Fpad] = t;
        }
        __syncthreads();
    }
}

// fam[a][f] = -(sum over chunks of the partial SSDs); per-block maxima as in k_combine.
__global__ void __launch_bounds__(256)
k_combine_f32(const double* __restrict__ part, double* __restrict__ fam, unsigned long long* __restrict__ blockmax,
              StepState* __restrict__ st, LibCfg c, int nchunk, int APAD, int n_agents) {
    __shared__ unsigned long long wmax[4];
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y;
    if (blockIdx.x == 0 && blockIdx.y == 0) reset_step_state(st, threadIdx.x, n_agents);
    unsigned long long key = 0;
    if (f < c.F) {
        double ssd = 0.0;
        for (int ch = 0; ch < nchunk; ++ch) ssd += part[((long long)ch * APAD + a) * c.Fpad + f];
        const double val = -ssd;
        fam[(long long)a * c.Fpad + f] = val;
        key = ordered_key(val);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = wmax[0];
        for (int i = 1; i < 4; ++i) m = wmax[i] > m ? wmax[i] : m;
        blockmax[(long long)a * gridDim.x + blockIdx.x] = m;
    }
}

// ------------------------------------------------------------------ ssd_f32 on the matrix cores (round 4)
// k_ssd_tiles computes (l - p)^2 directly: two vector operations per pixel and heading, which bound it by its instruction
// stream at 0.50-0.59 of the HBM peak.  The expansion  sum (l - p)^2 = N_l + N_p - 2 sum l p  turns the library stream into
// the B operand of an fp32 matrix product -- v_mfma_f32_32x32x1_2b_f32 (32 headings x 64 views per instruction and pixel) or
// v_mfma_f32_16x16x1_4b_f32 (16 headings x 64 views): with K = 1 and 2 / 4 blocks the 64 lanes of the B operand are 64
// DIFFERENT views, i.e. exactly the lane <-> view tiles the library already has (ftiles), one VGPR of a loaded float4 per
// instruction; the A operand is the patch pixel of heading lane & 31 (& 15), the same in every block.  Lane maps measured with
// tools/exp/mfma_f32_blocks.hip: result register r of lane l holds
//     32x32x1_2b: heading (r & 3) + 8 ((r & 15) >> 2) + 4 (l >> 5), view 32 (r >> 4) + (l & 31)
//     16x16x1_4b: heading (r & 3) + 4 (l >> 4),                     view 16 (r >> 2) + (l & 15)
// and both issue at 32 multiply-adds per cycle and SIMD (64.7 / 33.1 cycles): 500 000 views x 128x128 x 32 headings need 3.3 ms of
// matrix pipe at 2.4 GHz against 5.2 ms of HBM stream, 50 000 x 64x64 x 16 headings 42 us against 130.
// The expansion cancels -- a near match's small SSD is the difference of large numbers -- so these sums only SELECT: an fp32
// chain of 1024 pixels (then folded into a double) is off by at most gamma = 1024 u / (1 - 1024 u), u = 2^-24, times sum |l p| <=
// (N_l + N_p) / 2, hence every score lies within E = gamma (N_l + N_p) of its approximation; k_cand_f32x lists, per heading,
// every view whose interval reaches the best lower bound, k_resolve_f32 re-scores the listed pairs in the reference's own
// sequential double arithmetic (util.pyx:180-182) and k_decide takes minima and the decision from those exact values: every
// reported per-heading minimum is then the reference's double bit for bit, not merely within 1e-6.
constexpr int kF32xFold = 256;                         // q-steps (4 pixels each) per fp32 chain
constexpr double kF32xKappa = 6.2e-5;                  // fp32 form: >= gamma_1024 = 6.1039e-5 with room for the norms' own rounding
// two-term bf16 form (k_ssd_f32_bf16x2): l p is taken as lh ph + lh pl + ll ph with lh = bf16(l), ll = bf16(l - lh) -- what is
// left out (ll pl and the two residuals below 2^-18) is under 1.2e-5 |l p|; a chain of 256 pixels = 48 instructions of 16 products
// each, priced at 17 fp32 additions of error 2u apiece whatever the pipe's internal order or rounding: 9.8e-5
constexpr double kF32xKappaBf16 = 1.2e-4;
constexpr int kF32xShards = 32;                        // copies of a heading's lower bound the blocks of k_combine_f32x spread their atomics over

// N_f of every view, in double.  One thread per view (lane <-> view: coalesced), grid = Fpad / 64 blocks of 64.
__global__ void __launch_bounds__(64)
k_norm_f32(const float4* __restrict__ ftiles, double* __restrict__ vnorm, LibCfg c) {
    const long long g = blockIdx.x;
    const int lane = threadIdx.x;
    const float4* base = ftiles + g * c.gstride + lane;
    double n = 0.0;
    for (int q = 0; q < c.Q; ++q) {
        const float4 L = base[(long long)q * 64];
        n += (double)L.x * (double)L.x; n += (double)L.y * (double)L.y; n += (double)L.z * (double)L.z; n += (double)L.w * (double)L.w;
    }
    vnorm[g * 64 + lane] = n;
}

// Everything the matrix-core form needs of the A resident patches, one launch: blocks [0, nrow) write the A operand rows
// pprep[q][a] = float4 of patch a's pixels 4q .. 4q+3 (zero past the last heading / pixel), a < APAD; block nrow + a sums N_a of
// heading a's patch in double (a fixed-order sum: the same value every run) and zeroes the heading's lower bound for k_combine_f32x.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
// l as two bf16 terms: hi = bf16(l), lo = bf16(l - hi), each pair of pixels packed low half first
__device__ __forceinline__ void split_bf16(float x0, float x1, unsigned& hi, unsigned& lo) {
    const bf16x2_t h = {(__bf16)x0, (__bf16)x1};
    hi = __builtin_bit_cast(unsigned, h);
    const float h0 = __uint_as_float(hi << 16), h1 = __uint_as_float(hi & 0xffff0000u);
    const bf16x2_t l = {(__bf16)(x0 - h0), (__bf16)(x1 - h1)};
    lo = __builtin_bit_cast(unsigned, l);
}
// The A operand rows of the two-term bf16 form (k_ssd_f32_bf16x2): pprepb[pass][block of 16 pixels][hi / lo][lane] = the eight bf16
// of heading 32 pass + (lane & 31) at pixels 16 block + 8 (lane >> 5) + 0..7 -- v_mfma_f32_32x32x16_bf16's A map, lane for lane.
__global__ void __launch_bounds__(256)
k_prep_f32b(const float* __restrict__ raw, uint4* __restrict__ pprepb, LibCfg c, int A, int passes) {
    const int nb16 = (c.P + 15) / 16;
    const long long total = (long long)passes * nb16 * 64;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const int b = (int)((t >> 6) % nb16);
    const int pass = (int)((t >> 6) / nb16);
    const int a = pass * 32 + (lane & 31);
    unsigned hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int px = 16 * b + 8 * (lane >> 5) + 2 * i;
        const float x0 = (a < A && px < c.P) ? raw[(long long)a * c.P + px] : 0.f;
        const float x1 = (a < A && px + 1 < c.P) ? raw[(long long)a * c.P + px + 1] : 0.f;
        split_bf16(x0, x1, hi[i], lo[i]);
    }
    uint4* dst = pprepb + (((long long)pass * nb16 + b) * 2) * 64 + lane;
    dst[0] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    dst[64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

__global__ void __launch_bounds__(256)
k_prep_f32x(const float* __restrict__ raw, float4* __restrict__ pprep, double* __restrict__ pnorm, unsigned long long* __restrict__ lower,
            LibCfg c, int A, int APAD, int nrow) {
    if ((int)blockIdx.x >= nrow) {
        __shared__ double red[256];
        const int a = blockIdx.x - nrow;
        const float* pa = raw + (long long)a * c.P;
        double n = 0.0;
        for (int px = threadIdx.x; px < c.P; px += 256) n += (double)pa[px] * (double)pa[px];
        red[threadIdx.x] = n;
        __syncthreads();
        for (int s2 = 128; s2 > 0; s2 >>= 1) {
            if (threadIdx.x < s2) red[threadIdx.x] += red[threadIdx.x + s2];
            __syncthreads();
        }
        if (threadIdx.x == 0) pnorm[a] = red[0];
        if (threadIdx.x < kF32xShards) lower[a * kF32xShards + threadIdx.x] = 0ull;
        return;
    }
    const long long total = (long long)c.Q * APAD;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int a = (int)(t % APAD);
    const int q = (int)(t / APAD);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (a < A)
        for (int i = 0; i < 4; ++i) { const int px = 4 * q + i; if (px < c.P) v[i] = raw[(long long)a * c.P + px]; }
    pprep[t] = make_float4(v[0], v[1], v[2], v[3]);
}

typedef float v32f_t __attribute__((ext_vector_type(32)));
typedef float v16f32_t __attribute__((ext_vector_type(16)));
typedef float v4f_t __attribute__((ext_vector_type(4)));

// The cross terms sum l p of HB headings (at a_off) against one view group of 64 per item = (pixel chunk, view group); one wave
// per item, D q-steps of library AND patch rows in flight (the patch rows come out of L1 / L2: every wave of the chip reads the
// same ones), four MFMAs per q-step.  part[chunk][heading][view] (double) takes the item's sums.
// Measured (round 4, one MI355X): 50 000 views x 64x64 x 16 headings (16-wide instruction, D = 8, three waves per SIMD) 138-148 us =
// 5.5-5.9 TB/s, where k_ssd_tiles takes 186-205; 500 000 views x 128x128 x 32 headings (32-wide, D = 16, two waves per SIMD) 6.6 ms =
// 5.0 TB/s against 14.5 ms in two direct passes.  At 32 headings the pass is no longer bound by the stream: the matrix pipe is busy
// 67 % of the kernel at the 1.79 GHz the chip holds under it (PMC: SQ_VALU_MFMA_BUSY_CYCLES 8.0e6 of 1.2e7 cycles per SIMD, waves
// 77 % in issue stalls, 10 % in memory waits); D = 8 6.8 ms, three waves per SIMD with fp32 second-level sums 7.2-7.7 ms.
template <int HB, int D>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu((HB == 32 || D > 8) ? 2 : 3)))
k_ssd_f32_mfma(const float4* __restrict__ ftiles, const float4* __restrict__ pprep, double* __restrict__ part, LibCfg c, int nchunk,
               int apad_total, int a_off) {
    constexpr int R = HB == 32 ? 32 : 16;              // result registers per lane
    // D: q-steps (1 KB of library + 0.5 KB of patch rows each) in flight per wave
    static_assert(kF32xFold % D == 0, "the chains are folded on a step boundary");
    using acc_t = typename std::conditional<HB == 32, v32f_t, v16f32_t>::type;
    const int lane = threadIdx.x;
    const long long G = c.Fpad / 64;
    const long long n_items = G * nchunk;
    const int Q = c.Q;
    const int rows = (apad_total - a_off) < HB ? (apad_total - a_off) : HB;
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ch = (int)(item / G);
        const long long g = item - (long long)ch * G;
        const int q0 = (int)(((long long)ch * Q) / nchunk), q1 = (int)(((long long)(ch + 1) * Q) / nchunk);
        const v4f_t* lb = reinterpret_cast<const v4f_t*>(ftiles + g * c.gstride + lane);
        const v4f_t* pb = reinterpret_cast<const v4f_t*>(pprep + a_off + (lane & (HB - 1)));
        acc_t acc;
        double accd[R];
#pragma unroll
        for (int r = 0; r < R; ++r) { acc[r] = 0.f; accd[r] = 0.0; }
        v4f_t L[D], Pt[D];
        auto fetch = [&](int d, int q) {
            const int qq = q < q1 ? q : q1 - 1;                     // (past the chunk: its last rows again, not multiplied)
            L[d] = __builtin_nontemporal_load(lb + (long long)qq * 64);
            Pt[d] = pb[(long long)qq * apad_total];
        };
#pragma unroll
        for (int d = 0; d < D; ++d) fetch(d, q0 + d);
        int chain = 0;
        for (int q = q0; q < q1; q += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const v4f_t l = L[d], p = Pt[d];
                fetch(d, q + d + D);
                if (q + d < q1) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (HB == 32) acc = __builtin_amdgcn_mfma_f32_32x32x1f32(p[j], l[j], acc, 0, 0, 0);
                        else acc = __builtin_amdgcn_mfma_f32_16x16x1f32(p[j], l[j], acc, 0, 0, 0);
                    }
                }
            }
            chain += D;
            if (chain >= kF32xFold) {                               // the fp32 chains end here: into the doubles
#pragma unroll
                for (int r = 0; r < R; ++r) { accd[r] += (double)acc[r]; acc[r] = 0.f; }
                chain = 0;
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) accd[r] += (double)acc[r];
        double* dst = part + ((long long)ch * apad_total + a_off) * c.Fpad + g * 64;
        int ln = lane;                                              // opaque: the row addresses are not hoisted out of the item loop
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int hd = HB == 32 ? (r & 3) + 8 * ((r & 15) >> 2) + 4 * (ln >> 5) : (r & 3) + 4 * (ln >> 4);
            const int view = HB == 32 ? 32 * (r >> 4) + (ln & 31) : 16 * (r >> 2) + (ln & 15);
            if (hd < rows) dst[(long long)hd * c.Fpad + view] = accd[r];
        }
    }
}

// The same cross terms for a pass of 32 headings in the two-term bf16 form: l p = lh ph + lh pl + ll ph on v_mfma_f32_32x32x16_bf16
// (16x the fp32 instruction's rate).  The library stays fp32 in HBM (the exact resolver needs it); a wave turns each 16 pixels of its
// 64 views into bf16 pairs in registers (split_bf16) and into the instruction's B operand -- column = view & 31, k = 8 (lane >> 5) +
// 0..7 -- with four v_permlane32_swap per term: the upper lanes' low-k registers change places with the lower lanes' high-k ones,
// which leaves one register set holding views 0-31 and the other views 32-63.  Six instructions per 16 pixels and 64 views (192 matrix
// cycles where the fp32 form takes 1024) beside ~60 vector instructions: the pass is bound by its stream again.
template <int D>                               // blocks of 16 pixels in flight (4 KB of library + 2 KB of patch rows each)
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2)))
k_ssd_f32_bf16x2(const float4* __restrict__ ftiles, const uint4* __restrict__ pprepb, double* __restrict__ part, LibCfg c, int nchunk,
                 int apad_total, int a_off) {
    constexpr int kFoldBlocks = 16;            // 256 pixels per fp32 chain
    const int lane = threadIdx.x;
    const long long G = c.Fpad / 64;
    const long long n_items = G * nchunk;
    const int Q = c.Q;
    const int nb16 = (c.P + 15) / 16;
    const int rows = (apad_total - a_off) < 32 ? (apad_total - a_off) : 32;
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ch = (int)(item / G);
        const long long g = item - (long long)ch * G;
        const int b0 = (int)(((long long)ch * nb16) / nchunk), b1 = (int)(((long long)(ch + 1) * nb16) / nchunk);
        const v4f_t* lb = reinterpret_cast<const v4f_t*>(ftiles + g * c.gstride + lane);
        const v4u_t* pb = reinterpret_cast<const v4u_t*>(pprepb + ((long long)(a_off / 32) * nb16) * 128 + lane);
        v16f32_t acc0, acc1;
        double accd[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int r = 0; r < 32; ++r) accd[r] = 0.0;
        v4f_t L[D][4];
        v4u_t Ph[D], Pl[D];
        auto fetch_lib = [&](int d, int b) {
            const int bb = b < b1 ? b : b1 - 1;                     // (past the chunk: its last block again, not multiplied)
#pragma unroll
            for (int s2 = 0; s2 < 4; ++s2) {
                const int q = 4 * bb + s2;
                L[d][s2] = __builtin_nontemporal_load(lb + (long long)(q < Q ? q : Q - 1) * 64);
            }
        };
        auto fetch_rows = [&](int d, int b) {
            const int bb = b < b1 ? b : b1 - 1;
            Ph[d] = pb[(long long)bb * 128];
            Pl[d] = pb[(long long)bb * 128 + 64];
        };
#pragma unroll
        for (int d = 0; d < D; ++d) { fetch_lib(d, b0 + d); fetch_rows(d, b0 + d); }
        int chain = 0;
        for (int b = b0; b < b1; b += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                // the block's 16 pixels as bf16 pairs first: the float registers are then free for the fetch of block b + d + D
                unsigned H[8], Lo[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool real = 4 * (b + d) + (j >> 1) < Q;    // (q-steps past the last pixel: the clamped load is not theirs)
                    const float x0 = real ? L[d][j >> 1][(j & 1) * 2] : 0.f, x1 = real ? L[d][j >> 1][(j & 1) * 2 + 1] : 0.f;
                    split_bf16(x0, x1, H[j], Lo[j]);
                }
                fetch_lib(d, b + d + D);
                if (b + d < b1) {
                    // registers j (pixels 2j, 2j+1) and 4 + j (pixels 8 + 2j ..): upper lanes of the first <-> lower lanes of the second
                    unsigned hx[4], hy[4], lx[4], ly[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const auto sh = __builtin_amdgcn_permlane32_swap(H[j], H[4 + j], false, false);
                        hx[j] = sh[0]; hy[j] = sh[1];
                        const auto sl = __builtin_amdgcn_permlane32_swap(Lo[j], Lo[4 + j], false, false);
                        lx[j] = sl[0]; ly[j] = sl[1];
                    }
                    const bf16x8_t Ahi = __builtin_bit_cast(bf16x8_t, Ph[d]), Alo = __builtin_bit_cast(bf16x8_t, Pl[d]);
                    const bf16x8_t Bhx = __builtin_bit_cast(bf16x8_t, v4u_t{hx[0], hx[1], hx[2], hx[3]});
                    const bf16x8_t Bhy = __builtin_bit_cast(bf16x8_t, v4u_t{hy[0], hy[1], hy[2], hy[3]});
                    const bf16x8_t Blx = __builtin_bit_cast(bf16x8_t, v4u_t{lx[0], lx[1], lx[2], lx[3]});
                    const bf16x8_t Bly = __builtin_bit_cast(bf16x8_t, v4u_t{ly[0], ly[1], ly[2], ly[3]});
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ahi, Bhx, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ahi, Bhy, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Alo, Bhx, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Alo, Bhy, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ahi, Blx, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ahi, Bly, acc1, 0, 0, 0);
                }
                fetch_rows(d, b + d + D);
            }
            chain += D;
            if (chain >= kFoldBlocks) {                             // the fp32 chains end here: into the doubles
#pragma unroll
                for (int r = 0; r < 16; ++r) { accd[r] += (double)acc0[r]; acc0[r] = 0.f; accd[16 + r] += (double)acc1[r]; acc1[r] = 0.f; }
                chain = 0;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { accd[r] += (double)acc0[r]; accd[16 + r] += (double)acc1[r]; }
        double* dst = part + ((long long)ch * apad_total + a_off) * c.Fpad + g * 64;
        int ln = lane;                                              // opaque: the row addresses are not hoisted out of the item loop
        asm volatile("" : "+v"(ln));
#pragma unroll
        for (int r = 0; r < 32; ++r) {                              // the 32-wide fp32 form's map: register r of set r >> 4
            const int hd = (r & 3) + 8 * ((r & 15) >> 2) + 4 * (ln >> 5);
            const int view = 32 * (r >> 4) + (ln & 31);
            if (hd < rows) dst[(long long)hd * c.Fpad + view] = accd[r];
        }
    }
}

// fam[a][f] = -(N_f + N_a - 2 sum over chunks of part): the approximate SSD, negated like every score here; per block the
// largest LOWER bound fam - E (E = kappa (N_f + N_a)) as an ordered key.  grid = (ceil(Fpad / 256), A).
__global__ void __launch_bounds__(256)
k_combine_f32x(const double* __restrict__ part, const double* __restrict__ vnorm, const double* __restrict__ pnorm, double* __restrict__ fam,
               unsigned long long* __restrict__ lower, StepState* __restrict__ st, LibCfg c, int nchunk, int APAD, int n_agents, double kappa) {
    __shared__ unsigned long long wmax[4];
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y;
    if (blockIdx.x == 0 && blockIdx.y == 0) reset_step_state(st, threadIdx.x, n_agents);
    unsigned long long key = 0;
    if (f < c.F) {
        double dot = 0.0;
        for (int ch = 0; ch < nchunk; ++ch) dot += part[((long long)ch * APAD + a) * c.Fpad + f];
        const double n2 = vnorm[f] + pnorm[a];
        const double val = -(n2 - 2.0 * dot);
        fam[(long long)a * c.Fpad + f] = val;
        key = ordered_key(val - kappa * n2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = wmax[0];
        for (int i = 1; i < 4; ++i) m = wmax[i] > m ? wmax[i] : m;
        // (zeroed by k_prep_f32x with the patches this step scores; 32 shards per heading: a few hundred blocks' atomics on ONE word
        // serialise at ~100 ns each -- 21 of this kernel's 32 us at 50 000 views x 16 headings)
        atomicMax(&lower[a * kF32xShards + (blockIdx.x & (kF32xShards - 1))], m);
    }
}

// Per heading: every view whose interval [fam - E, fam + E] reaches the heading's best lower bound goes to the candidate list
// (cand[i] = heading << 40 | view; st->ncand counts, possibly past kCandCap: the step is then redone exactly).  Block 0 leaves the
// lower bounds as the step's approximate per-heading maxima.  grid = ceil(F / 256).
__global__ void __launch_bounds__(256)
k_cand_f32x(const double* __restrict__ fam, const unsigned long long* __restrict__ lower, const double* __restrict__ vnorm,
            const double* __restrict__ pnorm, StepState* __restrict__ st, unsigned long long* __restrict__ cand, LibCfg c, int A, double kappa) {
    __shared__ unsigned long long s_lb[kMaxHeadings];
    if (threadIdx.x < kMaxHeadings) s_lb[threadIdx.x] = 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < A * kF32xShards; i += blockDim.x) atomicMax(&s_lb[i / kF32xShards], lower[i]);
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x < A) {
        __hip_atomic_store(&st->amax[threadIdx.x], s_lb[threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&st->aview[threadIdx.x], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= c.F) return;
    const double nf = vnorm[f];
    for (int a = 0; a < A; ++a) {
        const double v = fam[(long long)a * c.Fpad + f];
        const double e = kappa * (nf + pnorm[a]);
        if (v + e >= key_to_double(s_lb[a])) {
            const unsigned long long pos = atomicAdd(&st->ncand, 1ull);
            if (pos < (unsigned long long)kCandCap) cand[pos] = ((unsigned long long)a << 40) | (unsigned long long)f;
        }
    }
}

// ------------------------------------------------------------------ ssd_u8 metric: exact SSD of uint8 views on the int8 matrix cores
// The same `ssds` (navsim/util.pyx:171-184: sum over i,j of (a[i,j] - b[i,j])**2) for single-channel uint8 views.  Squared
// differences ARE bilinear once expanded, and in integers the expansion loses nothing:
//     sum (a - b)^2 = sum a'^2 + sum b'^2 - 2 sum a' b',     a' = a - 128, b' = b - 128 in [-128, 127]
// -- the cross term is an int8 GEMM (headings x views x pixels, int32 accumulate: |sum| <= P * 2^14, P <= 131 071), the norms are
// per-view and per-heading constants.  Every score is the exact integer the reference's float64 loop produces for uint8 inputs
// (all of its partial sums are integers below 2^53), so ties are decided by index (k_tail's exact rule) and nothing is re-scored.
// Layouts (one K-step = 32 pixels = one v_mfma_i32_32x32x32_i8):
//   u8tiles[g][k][lane] : uint4 = 16 pixels 32 k + 16 (lane >> 5) .. of view 32 g + (lane & 31), bytes x ^ 0x80 (= x - 128 as an
//                         int8); bytes past the last pixel or view are 0 (they add nothing to any sum).  (View groups an odd number
//                         of rows apart instead of K measured the same: 200 000 views x 128x128, 634 against 642 us);
//   u8prep[pass][k][lane] : the same of heading 32 pass + (lane & 31): the A operand of a pass over the library;
//   vnorm[f], pnorm[a] : sum of (x - 128)^2.
// One byte per pixel crosses HBM per pass of 32 headings; the kernel is bound by that stream (one MFMA per KB of library).
__global__ void __launch_bounds__(256)
k_retile_u8(const unsigned char* __restrict__ raw, uint4* __restrict__ tiles, unsigned long long* __restrict__ vnorm, LibCfg c, int K) {
    const long long total = (c.Fpad / 32) * (long long)K * 64;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const long long r = t >> 6;
    const int k = (int)(r % K);
    const long long f = (r / K) * 32 + (lane & 31);
    unsigned w[4] = {0u, 0u, 0u, 0u};
    unsigned long long nrm = 0;
    if (f < c.F) {
        const int px0 = 32 * k + 16 * (lane >> 5);
        for (int i = 0; i < 16; ++i) {
            if (px0 + i < c.P) {
                const unsigned x = raw[f * (long long)c.P + px0 + i];
                const int d = (int)x - 128;
                nrm += (unsigned long long)(d * d);
                w[i >> 2] |= (x ^ 0x80u) << (8 * (i & 3));
            }
        }
        if (nrm) atomicAdd(&vnorm[f], nrm);
    }
    tiles[t] = make_uint4(w[0], w[1], w[2], w[3]);
}

// One byte of every sensed HSV pixel (the channel the ssd_u8 plug-in compares): uint8[n_px][3] -> uint8[n_px].
__global__ void __launch_bounds__(256)
k_take_channel(const unsigned char* __restrict__ hsv, unsigned char* __restrict__ out, long long n_px, int channel) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_px) out[t] = hsv[t * 3 + channel];
}

// u8prep and pnorm of the resident patches (pnorm zeroed by the caller).  grid = ceil(passes * K * 64 / 256).
__global__ void __launch_bounds__(256)
k_prep_u8(const unsigned char* __restrict__ raw, uint4* __restrict__ prep, unsigned long long* __restrict__ pnorm, LibCfg c, int K, int A, int passes) {
    // (the norms meet in LDS first: one atomic per heading and block -- a thread's own atomic on its heading's word queued 256 deep
    // behind the others' at 64x64 x 16 headings, half of this kernel's 7.7 us)
    __shared__ unsigned long long s_nrm[kMaxHeadings];
    if (threadIdx.x < kMaxHeadings) s_nrm[threadIdx.x] = 0;
    __syncthreads();
    const long long total = (long long)passes * K * 64;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < total) {
        const int lane = (int)(t & 63);
        const long long r = t >> 6;
        const int k = (int)(r % K);
        const int a = (int)(r / K) * 32 + (lane & 31);
        unsigned w[4] = {0u, 0u, 0u, 0u};
        unsigned long long nrm = 0;
        if (a < A) {
            const int px0 = 32 * k + 16 * (lane >> 5);
            for (int i = 0; i < 16; ++i) {
                if (px0 + i < c.P) {
                    const unsigned x = raw[(long long)a * c.P + px0 + i];
                    const int d = (int)x - 128;
                    nrm += (unsigned long long)(d * d);
                    w[i >> 2] |= (x ^ 0x80u) << (8 * (i & 3));
                }
            }
            if (nrm) atomicAdd(&s_nrm[a], nrm);
        }
        prep[t] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    if (threadIdx.x < kMaxHeadings && s_nrm[threadIdx.x]) atomicAdd(&pnorm[threadIdx.x], s_nrm[threadIdx.x]);
}

// The cross terms of one pass (32 headings at a_off) over the library.  A workgroup of 8 waves keeps the pass's operand rows of a
// chunk of KC K-steps in LDS (KC KB) and walks its items -- the library cut evenly into n_items ranges of at most 8 TL view groups,
// TL per wave (nothing but the stream sets this kernel's time, so every CU should stream the same share) -- through that chunk
// before it loads the next: the library crosses HBM once per pass, and part[chunk][heading][view] (int32) takes the chunk's sums.
// A wave streams its view groups' rows (1 KB per wave-instruction, consecutive K-steps consecutive in memory) into registers eight
// K-steps ahead: two sets of TL x 8 rows alternate, so 16 KB per wave are on their way while the other 16 KB are multiplied.
template <int TL>
__global__ void __launch_bounds__(512)
k_ssd_u8_mfma(const uint4* __restrict__ tiles, const uint4* __restrict__ prep, int* __restrict__ part, LibCfg c, int K, int KC, int nchunk,
              int apad_total, int a_off, long long n_items) {
    extern __shared__ uint4 lds_rows[];                       // [KC][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long G32 = c.Fpad / 32;
    const int rows = (apad_total - a_off) < 32 ? (apad_total - a_off) : 32;
    for (int ch = 0; ch < nchunk; ++ch) {
        const int k0 = ch * KC, kn = (K - k0) < KC ? (K - k0) : KC;
        const uint4* base[TL];
        long long gidx[TL];
        bool live[TL];
        v4i_t buf[2][TL][8];
        auto aim = [&](long long item) {                       // this wave's view groups of an item
            const long long g0 = (item * G32) / n_items, g1 = ((item + 1) * G32) / n_items;
#pragma unroll
            for (int t = 0; t < TL; ++t) {
                const long long g = g0 + wave * TL + t;
                live[t] = g < g1;
                gidx[t] = live[t] ? g : G32 - 1;               // a slot without a view group re-reads the last one (never stored)
                base[t] = tiles + (gidx[t] * K + k0) * 64 + lane;
            }
        };
        auto load = [&](v4i_t (&dst)[TL][8], int kb) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = kb + j < kn ? kb + j : kn - 1;   // (past the chunk: its last row again, not multiplied)
#pragma unroll
                for (int t = 0; t < TL; ++t) dst[t][j] = __builtin_nontemporal_load(reinterpret_cast<const v4i_t*>(base[t] + (long long)k * 64));
            }
        };
        // the first item's first rows are asked for BEFORE the operand rows are copied into LDS: the two round trips overlap (the copy
        // and its barriers used to stand in front of the stream: ~4 of the kernel's 39 us at 50 000 views x 64x64)
        bool first = (long long)blockIdx.x < n_items;
        if (first) { aim(blockIdx.x); load(buf[0], 0); }
        __syncthreads();                                       // everybody is done with the chunk before
        for (int i = tid; i < kn * 64; i += 512) lds_rows[i] = prep[(long long)k0 * 64 + i];
        __syncthreads();
        for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
            if (!first) { aim(item); load(buf[0], 0); }
            first = false;
            v16i_t acc[TL];
#pragma unroll
            for (int t = 0; t < TL; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0;
            auto multiply = [&](const v4i_t (&src)[TL][8], int kb) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (kb + j < kn) {
                        const v4i_t a = *reinterpret_cast<const v4i_t*>(&lds_rows[(kb + j) * 64 + lane]);
#pragma unroll
                        for (int t = 0; t < TL; ++t) acc[t] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, src[t][j], acc[t], 0, 0, 0);      // headings x views
                    }
                }
            };
            for (int kb = 0; kb < kn; kb += 16) {
                if (kb + 8 < kn) load(buf[1], kb + 8);
                multiply(buf[0], kb);
                if (kb + 16 < kn) load(buf[0], kb + 16);
                if (kb + 8 < kn) multiply(buf[1], kb + 8);
            }
#pragma unroll
            for (int t = 0; t < TL; ++t) {
                if (!live[t]) continue;
                int* dst = part + ((long long)ch * apad_total + a_off) * c.Fpad + gidx[t] * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    if (m < rows) dst[(long long)m * c.Fpad] = acc[t][r];
                }
            }
        }
    }
}

// fam[a][f] = -(vnorm[f] + pnorm[a] - 2 * sum over chunks of part): the exact SSD, negated like ssd_f32's; per-block maxima as in k_combine.
__global__ void __launch_bounds__(256)
k_combine_u8(const int* __restrict__ part, const unsigned long long* __restrict__ vnorm, const unsigned long long* __restrict__ pnorm,
             double* __restrict__ fam, unsigned long long* __restrict__ blockmax, StepState* __restrict__ st, LibCfg c, int nchunk, int APAD,
             int n_agents) {
    __shared__ unsigned long long wmax[4];
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y;
    if (blockIdx.x == 0 && blockIdx.y == 0) reset_step_state(st, threadIdx.x, n_agents);
    unsigned long long key = 0;
    if (f < c.F) {
        long long dot = 0;
        for (int ch = 0; ch < nchunk; ++ch) dot += (long long)part[((long long)ch * APAD + a) * c.Fpad + f];
        const long long ssd = (long long)vnorm[f] + (long long)pnorm[a] - 2 * dot;
        const double val = -(double)ssd;
        fam[(long long)a * c.Fpad + f] = val;
        key = ordered_key(val);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = wmax[0];
        for (int i = 1; i < 4; ++i) m = wmax[i] > m ? wmax[i] : m;
        blockmax[(long long)a * gridDim.x + blockIdx.x] = m;
    }
}

// Every (heading, view) SSD exact (overflow fallback / exact mode of the f32 metric).  grid = (G, ceil(A/4)), block (64,4).
__global__ void __launch_bounds__(256)
k_exact_all_f32(const float4* __restrict__ ftiles, const float* __restrict__ raw_patches, double* __restrict__ fam,
                unsigned long long* __restrict__ groupmax, StepState* __restrict__ st, LibCfg c, int A, int n_agents) {
    const int lane = threadIdx.x;
    if (blockIdx.x == 0 && blockIdx.y == 0) reset_step_state(st, threadIdx.y * 64 + threadIdx.x, n_agents);
    const int a = blockIdx.y * 4 + threadIdx.y;
    if (a >= A) return;
    const long long g = blockIdx.x;
    const long long f = g * 64 + lane;
    const float4* base = ftiles + g * c.gstride + lane;
    const float* pa = raw_patches + (long long)a * c.P;
    double diff = 0.0;
    for (int q = 0; q < c.Q; ++q) {
        const float4 L = base[(long long)q * 64];
        const float lw[4] = {L.x, L.y, L.z, L.w};
        for (int i = 0; i < 4; ++i) {
            const int px = q * 4 + i;
            if (px >= c.P) break;
            const double d = (double)pa[px] - (double)lw[i];
            diff += d * d;
        }
    }
    const double val = -diff;
    unsigned long long key = 0;
    if (f < c.F) {
        fam[(long long)a * c.Fpad + f] = val;
        key = ordered_key(val);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other > key ? other : key;
    }
    if (lane == 0) groupmax[(long long)a * gridDim.x + g] = key;
}

// Exact SSD of listed (heading, view) candidates: the reference's sequential double sum (util.pyx:180-182), negated.
__global__ void __launch_bounds__(64)
k_resolve_f32(const float4* __restrict__ ftiles, const float* __restrict__ raw_patches, const StepState* __restrict__ st,
              const unsigned long long* __restrict__ cand, double* __restrict__ cand_exact, LibCfg c) {
    __shared__ double terms[256];
    const unsigned long long n_all = st->ncand;
    if (n_all > (unsigned long long)kCandCap) return;
    const int n = (int)n_all;
    const int lane = threadIdx.x;
    for (int ci = blockIdx.x; ci < n; ci += gridDim.x) {
        const unsigned long long cf = cand[ci];
        const int a = (int)(cf >> 40);
        const long long f = (long long)(cf & 0xffffffffffull);
        const float4* base = ftiles + (f >> 6) * c.gstride + (f & 63);
        const float* pa = raw_patches + (long long)a * c.P;
        double diff = 0.0;
        // the next round's view and patch pixels are fetched while this round's 256 terms are summed (the sum is the serial part:
        // fetched in its own round, each of the 16 rounds of a 64x64 view paid a memory round trip on top -- 38 us per step)
        auto fetch = [&](int qb, float4& L, float (&pv)[4]) {
            const int q = qb + lane;
            L = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < c.Q) L = base[(long long)q * 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int px = q * 4 + i; pv[i] = (q < c.Q && px < c.P) ? pa[px] : 0.f; }
        };
        float4 Ln;
        float pn[4];
        fetch(0, Ln, pn);
        for (int qb = 0; qb < c.Q; qb += 64) {
            const int q = qb + lane;
            const float4 L = Ln;
            const float pv[4] = {pn[0], pn[1], pn[2], pn[3]};
            if (qb + 64 < c.Q) fetch(qb + 64, Ln, pn);
            if (q < c.Q) {
                const float lw[4] = {L.x, L.y, L.z, L.w};
                for (int i = 0; i < 4; ++i) {
                    const int px = q * 4 + i;
                    double t = 0.0;
                    if (px < c.P) { const double d = (double)pv[i] - (double)lw[i]; t = d * d; }
                    terms[lane * 4 + i] = t;
                }
            }
            __syncthreads();
            int npx = c.P - qb * 4;
            npx = npx > 256 ? 256 : npx;
            // the reference's sequential order; the LDS reads are batched so that only the adds are serial (as k_resolve does)
            int i = 0;
            for (; i + 16 <= npx; i += 16) {
                double t[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) t[k] = terms[i + k];
#pragma unroll
                for (int k = 0; k < 16; ++k) diff += t[k];
            }
            for (; i < npx; ++i) diff += terms[i];
            __syncthreads();
        }
        if (lane == 0) cand_exact[ci] = -diff;
    }
}

// ------------------------------------------------------------------ sensor model (the step either side of scoring)
// get_sensor_mat of the reference (navsim/NavBySceneFamiliarity.py:151-192) for n poses at once:
//   fill_sensor_from  (navsim/util.pyx:137-168): rotated nearest-neighbour crop, double arithmetic in the
//                     reference's operation order (no FMA contraction), C round() = half away from zero,
//                     landscape indexed [y, x]; negative indices wrap, indices past the end are an error;
//   downscale_chem    (navsim/util.pyx:91-134): per block V = round(mean V), H = hue with the largest summed
//                     saturation (lowest hue on ties, hue 0 when every sum is 0), S = ((sum / rows) * cols) & 0xFF
//                     -- C integer division then multiply (util.pyx:131), uint8 cast wraps;
//   level quantisation through float32 (:176-186): folded into a 256-entry table per channel made on the host;
//   mask of the middle columns (:189-190).
struct SensorCfg {
    int rows, cols;          // landscape
    int sw, sh;              // sensor pixels (sensor_dimensions = [w, h])
    int pw, ph;              // landscape pixels per sensor pixel (sensor_pixel_dimensions = [w, h])
    int mask_n;              // mask_middle_n
};
struct Pose { double x, y, c, s; };   // position and cos/sin of -(pi/2 - angle), computed by the host's libm

__device__ __forceinline__ bool sense_fetch(const unsigned char* __restrict__ land, const SensorCfg& g, const Pose& p,
                                            int i, int j, unsigned& H, unsigned& S, unsigned& V) {
    const double px = (double)j - 0.5 * (double)(g.sw * g.pw);
    const double py = (double)i - 0.5 * (double)(g.sh * g.ph);
    const double rx = px * p.c - py * p.s;
    const double ry = px * p.s + py * p.c;
    long long iy = (long long)round(ry + p.y);
    long long ix = (long long)round(rx + p.x);
    if (iy < 0) iy += g.rows;
    if (ix < 0) ix += g.cols;
    if (iy < 0 || ix < 0 || iy >= g.rows || ix >= g.cols) return false;
    const unsigned char* q = land + (iy * (long long)g.cols + ix) * 3;
    H = q[0]; S = q[1]; V = q[2];
    return true;
}

// One thread per sensor pixel of every pose.  out: uint8[n][sh][sw][3].
__global__ void k_sense(const unsigned char* __restrict__ land, const Pose* __restrict__ poses, int n, SensorCfg g,
                        const unsigned char* __restrict__ lut, unsigned char* __restrict__ out, int* __restrict__ err) {
    const long long total = (long long)n * g.sh * g.sw;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int bj = (int)(t % g.sw);
    const int bi = (int)((t / g.sw) % g.sh);
    const Pose p = poses[t / ((long long)g.sw * g.sh)];
    const int nblk = g.pw * g.ph;
    unsigned oh = 0, os = 0, ov = 0;
    bool ok = true;
    if (nblk == 1) {
        unsigned H, S, V;
        ok = sense_fetch(land, g, p, bi, bj, H, S, V);
        if (ok) { oh = S > 0 ? H : 0u; os = S; ov = V; }      // a zero-saturation pixel loses the argmax to hue 0
    } else {
        long long vsum = 0, best_sum = 0;
        unsigned best_hue = 0;
        for (int k = 0; k < nblk && ok; ++k) {
            unsigned Hk, Sk, Vk;
            ok = sense_fetch(land, g, p, bi * g.ph + k / g.pw, bj * g.pw + k % g.pw, Hk, Sk, Vk);
            if (!ok) break;
            vsum += Vk;
            long long sum = 0;                                  // summed saturation of pixel k's hue over the block
            for (int m = 0; m < nblk; ++m) {
                unsigned Hm, Sm, Vm;
                if (!sense_fetch(land, g, p, bi * g.ph + m / g.pw, bj * g.pw + m % g.pw, Hm, Sm, Vm)) { ok = false; break; }
                if (Hm == Hk) sum += Sm;
            }
            if (sum > best_sum || (sum == best_sum && sum > 0 && Hk < best_hue)) { best_sum = sum; best_hue = Hk; }
        }
        if (ok) {
            oh = best_sum > 0 ? best_hue : 0u;
            os = (unsigned)(((best_sum / g.ph) * g.pw) & 0xFF);
            ov = (unsigned)(long long)round((double)vsum / (double)nblk);
        }
    }
    if (!ok) { atomicOr(err, 1); return; }
    oh = lut[oh]; os = lut[256 + os]; ov = lut[512 + ov];
    const int mid = g.sw / 2;
    if (bj >= mid - g.mask_n && bj < mid + g.mask_n) { oh = 0; os = 0; ov = 0; }
    unsigned char* o = out + t * 3;
    o[0] = (unsigned char)oh; o[1] = (unsigned char)os; o[2] = (unsigned char)ov;
}

// One sensor pixel of one pose (k_patch_prep senses the headings' patches with it, four pixels per thread).
__device__ __forceinline__ bool sense_pixel(const unsigned char* __restrict__ land, const SensorCfg& g, const Pose& p,
                                            const unsigned char* __restrict__ lut, int bi, int bj,
                                            unsigned& oh, unsigned& os, unsigned& ov) {
    const int nblk = g.pw * g.ph;
    oh = os = ov = 0;
    if (nblk == 1) {
        unsigned H, S, V;
        if (!sense_fetch(land, g, p, bi, bj, H, S, V)) return false;
        oh = S > 0 ? H : 0u; os = S; ov = V;
    } else {
        long long vsum = 0, best_sum = 0;
        unsigned best_hue = 0;
        for (int k = 0; k < nblk; ++k) {
            unsigned Hk, Sk, Vk;
            if (!sense_fetch(land, g, p, bi * g.ph + k / g.pw, bj * g.pw + k % g.pw, Hk, Sk, Vk)) return false;
            vsum += Vk;
            long long sum = 0;
            for (int m = 0; m < nblk; ++m) {
                unsigned Hm, Sm, Vm;
                if (!sense_fetch(land, g, p, bi * g.ph + m / g.pw, bj * g.pw + m % g.pw, Hm, Sm, Vm)) return false;
                if (Hm == Hk) sum += Sm;
            }
            if (sum > best_sum || (sum == best_sum && sum > 0 && Hk < best_hue)) { best_sum = sum; best_hue = Hk; }
        }
        oh = best_sum > 0 ? best_hue : 0u;
        os = (unsigned)(((best_sum / g.ph) * g.pw) & 0xFF);
        ov = (unsigned)(long long)round((double)vsum / (double)nblk);
    }
    oh = lut[oh]; os = lut[256 + os]; ov = lut[512 + ov];
    const int mid = g.sw / 2;
    if (bj >= mid - g.mask_n && bj < mid + g.mask_n) { oh = 0; os = 0; ov = 0; }
    return true;
}

// ------------------------------------------------------------------ per-step patch preparation
// ONE kernel makes everything a step needs of its patches, whatever their source (MODE): uploaded raw bytes (0), the sensor
// model at the headings' poses (1: dv_sense_patches, dv_sense_step*), or the synthetic stream of the benchmarks (2).
// One block of 256 threads per (heading a, range of 256 sensor pixels); phase 1, one pixel per thread, leaves the pixels'
// raw bytes and stored-plane bytes in LDS; phase 2 writes from there
//   raw[a][P][3]            the patches' bytes (MODE 1 and 2; the exact kernels and the tie resolver read them);
//   prep[pl][q][j][APAD]    dword = 4 pixels (16q + 4j ..) of heading a in stored byte plane pl: the SGPR operands of v_sad_u8;
//   coef4 / coef            the coefficient images of the matrix-core kernel (layouts below, "bit-plane library"): 256 pixels are
//                           T whole K-steps of a segment with T planes per pixel, so a block owns whole 16-byte entries of both
//                           images -- one thread builds one entry from the bytes in LDS;
//   acc->hs[a]              byte path: sum over pixels whose hue is outside the library's hue set of S (+ the excess of a
//                           clamped signed-saturation byte), see plane_byte;
//   acc->bhs[a], bv[a]      bit-plane path: everything of the two sums that does not depend on the view.  With the
//                           thermometer identity the patch-only terms of one byte add up to
//                           (l_0 - a)+ + (a - l_max)+ + sum_t alpha_t = |a - l_0|, l_0 the plane's smallest library value;
//   acc->off                nonzero when some patch byte lies strictly inside a gap between two library levels: the scoring
//                           kernel then takes its int8 form (k_sad_mfma_dual);
//   acc->err                bit per agent of the pass: its sensor footprint reached past the end of the landscape.
// The sums are folded per block and added with one integer atomic each, so they must start at zero: `next` is the OTHER
// set of the pair, which nothing uses during this step -- block 0 clears it for the next preparation, and no memset
// sits on a step's path.  Image entries of K-steps past a segment's last pixel are never written: they were zeroed when the
// images were allocated.
// (the sums are read with acc_sum: kAccWays partial sums per heading, a cache line each -- with the 32 headings' sums side by side in
// one line, the 2048 blocks of a 128x128 x 32 heading preparation queued on three lines for most of the kernel's 28 us; a line per
// heading 16 us)
struct PrepAcc {
    int hs[kAccWays * kMaxHeadings * kAccStride];
    int bhs[kAccWays * kMaxHeadings * kAccStride];
    int bv[kAccWays * kMaxHeadings * kAccStride];
    unsigned long long err;
    unsigned off;
    unsigned pad;
};
constexpr int kPrepPlanes = 16;               // == kMaxBitPlanes (defined with the bit-plane library below)
struct PrepBits {
    int enabled;                              // the library has bit planes (build_bit_planes)
    int fp4;                                  // ... and an fp4 form: the E2M1 sign image is written too
    int T[2], NK[2];                          // planes per pixel and K-steps of the HS and the V segment
    unsigned tbl[kPrepPlanes];                // per bit plane: byte plane | lo << 8 | w << 16 | wfull << 24
    unsigned char lmin[kMaxHues + 1];         // smallest library value of each stored byte plane
    unsigned ok[kMaxHues + 1][8];             // bit v of plane pl: patch byte v has fp4 coefficients (on a level, or outside the range)
};
// The headings' poses travel as a kernel argument (64 x 32 bytes): no host-to-device copy on the step's path.
// ------------------------------------------------------------------ error / coverage metrics of the agent
// update_error of the reference (navsim/NavBySceneFamiliarity.py:252-276) for one position: the distance to every
// training point in the reference's double arithmetic (delta*delta summed, sqrt; no contraction), its minimum, and
// the coverage marks `dist <= reach` (the reference ORs them in only when the minimum is within reach, which is the
// same set: no distance is within reach unless the smallest is).  The last block to arrive hands {nearest, seq} to the
// host through mapped memory.  Off the step's critical path: the host collects the answer one step later.
struct PathErrState { unsigned long long minkey; unsigned ticket; unsigned pad; };
struct alignas(16) PathErrOut { double nearest; unsigned long long seq; };

// Block `blk` of `nblk` (256 threads each) of the computation: its own kernel (k_path_error), or the blocks behind the preparation
// blocks of k_patch_prep when an agent step asks for both in one launch (dv_agent_step).
struct PathErrArgs {
    const double* xy; long long n; double x, y, reach; unsigned char* cover; PathErrState* st; PathErrOut* out; unsigned long long seq;
    int nblk;                         // 0: nothing asked for
};
__device__ __forceinline__ void path_error_block(const PathErrArgs& pe, int blk) {
    const double* __restrict__ xy = pe.xy;
    const long long n = pe.n;
    const double x = pe.x, y = pe.y, reach = pe.reach;
    unsigned char* __restrict__ cover = pe.cover;
    PathErrState* __restrict__ st = pe.st;
    PathErrOut* __restrict__ out = pe.out;
    const unsigned long long seq = pe.seq;
    __shared__ unsigned long long wmin[4];
    unsigned long long key = ~0ull;
    for (long long i = (long long)blk * blockDim.x + threadIdx.x; i < n; i += (long long)pe.nblk * blockDim.x) {
        double dx = xy[2 * i] - x, dy = xy[2 * i + 1] - y;
        dx *= dx;
        dy *= dy;
        const double dist = sqrt(dx + dy);
        if (dist <= reach) cover[i] = 1;
        const unsigned long long k = (unsigned long long)__double_as_longlong(dist);     // dist >= 0: bit order = value order
        key = k < key ? k : key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other < key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) wmin[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = wmin[0];
        for (int i = 1; i < 4; ++i) m = wmin[i] < m ? wmin[i] : m;
        // The block's minimum and its arrival are agent-scope atomics, the minimum a RETURNING one waited for before the ticket is
        // drawn (performed at the device's coherence point by then); the last block reads the minimum with an agent-scope atomic load.
        // "8-byte agent atomics both sides" of MI355X_MICROARCH.md's valid hand-off forms -- the two __threadfence() that stood here
        // (an L2 write-back and an L1 invalidate each, in every block) cost the agent's step 3-4 us once these blocks ride in the
        // preparation launch (dv_agent_step); nothing else is handed over (the coverage marks are read after a stream wait).
        const unsigned long long before = __hip_atomic_fetch_min(&st->minkey, m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" :: "v"(before) : "memory");
        if (__hip_atomic_fetch_add(&st->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)pe.nblk - 1u) {
            const unsigned long long all = __hip_atomic_load(&st->minkey, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&st->minkey, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&st->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // answer and sequence number leave in ONE 16-byte store to the mapped host record (one PCIe write: the host,
            // which polls the sequence word and then reads the answer, can never pair a new number with an old answer),
            // so no system-scope fence sits between them
            v4u_t rec;
            rec.x = (unsigned)all; rec.y = (unsigned)(all >> 32); rec.z = (unsigned)seq; rec.w = (unsigned)(seq >> 32);
            *reinterpret_cast<v4u_t*>(out) = rec;
        }
    }
}

__global__ void __launch_bounds__(256)
k_path_error(PathErrArgs pe) { path_error_block(pe, (int)blockIdx.x); }

// update_error for the agents of an ensemble at once: block (b, j) walks its share of the training points for agent j -- the same
// arithmetic as path_error_block -- marks agent j's OWN coverage array (slot) and folds the minimum into minkey[j] with one atomic
// per block; the host reads the minima behind the kernel (dv_path_error_batch is synchronous: an ensemble step is ~1 ms).
constexpr int kPathBatch = 64;
struct PathBatchArgs {
    const double* xy; long long n; double reach; unsigned char* cover; long long cover_stride; unsigned long long* minkey;
    double x[kPathBatch], y[kPathBatch]; int slot[kPathBatch];
};
__global__ void __launch_bounds__(256)
k_path_error_batch(PathBatchArgs pa) {
    __shared__ unsigned long long wmin[4];
    const int j = blockIdx.y;
    const double x = pa.x[j], y = pa.y[j], reach = pa.reach;
    const double* __restrict__ xy = pa.xy;
    unsigned char* __restrict__ cover = pa.cover + (long long)pa.slot[j] * pa.cover_stride;
    unsigned long long key = ~0ull;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < pa.n; i += (long long)gridDim.x * blockDim.x) {
        double dx = xy[2 * i] - x, dy = xy[2 * i + 1] - y;
        dx *= dx;
        dy *= dy;
        const double dist = sqrt(dx + dy);
        if (dist <= reach) cover[i] = 1;
        const unsigned long long k = (unsigned long long)__double_as_longlong(dist);     // dist >= 0: bit order = value order
        key = k < key ? k : key;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o);
        key = other < key ? other : key;
    }
    if ((threadIdx.x & 63) == 0) wmin[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long m = wmin[0];
        for (int i = 1; i < 4; ++i) m = wmin[i] < m ? wmin[i] : m;
        atomicMin(&pa.minkey[j], m);
    }
}

struct PoseSet { Pose p[kMaxHeadings]; };

template <int MODE>
__global__ void __launch_bounds__(256)
k_patch_prep(const unsigned char* __restrict__ land, const PoseSet poses, int A, SensorCfg g, const unsigned char* __restrict__ lut,
             unsigned char* __restrict__ raw, unsigned* __restrict__ prep, LibCfg c, int APAD, PrepAcc* __restrict__ acc,
             PrepAcc* __restrict__ next, int A_agent, PrepBits pb, unsigned long long seed, uint4* __restrict__ coef,
             uint4* __restrict__ coef4, int what, PathErrArgs pe) {
    // blocks behind the preparation's own: the agent's error / coverage metrics of the position the last step ended at (dv_agent_step:
    // one launch for both -- a launch of its own cost the step ~4 us of host time, on a second stream more)
    if (pe.nblk > 0 && (int)blockIdx.x >= (int)gridDim.x - pe.nblk) { path_error_block(pe, (int)blockIdx.x - ((int)gridDim.x - pe.nblk)); return; }
    // what: bit 0 = the byte path's operand dwords (prep) -- 393 000 four-byte stores into lines shared by 32 headings at 128x128 x 32
    // headings, which a step on the matrix cores never reads: launch_patch_prep leaves them out there, and launch_int_scoring has
    // them written from the raw bytes (MODE 0, what = 1) should a byte kernel run on these patches after all;
    // bit 1 = everything else.
    const int tid = threadIdx.x;
    const bool all = (what & 2) != 0;
    if (blockIdx.x == 0 && all) {
        if (tid < kAccWays * kMaxHeadings) { next->hs[tid * kAccStride] = 0; next->bhs[tid * kAccStride] = 0; next->bv[tid * kAccStride] = 0; }
        if (tid == 0) { next->err = 0; next->off = 0; }
    }
    __shared__ unsigned char s_lut[768];                   // the sensor's level tables (3 x 256 bytes)
    __shared__ unsigned s_ok[(kMaxHues + 1) * 8];
    __shared__ unsigned s_tbl[kPrepPlanes];
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[256 * 3];
    __shared__ __attribute__((aligned(16))) unsigned char s_pl[(kMaxHues + 1)][256];      // stored-plane bytes of the block's pixels
    __shared__ int s_red[4][4];
    __shared__ __attribute__((aligned(16))) unsigned s_img4[kPrepPlanes * 32];      // the block's entries of the fp4 coefficient image ...
    __shared__ __attribute__((aligned(16))) unsigned s_img8[kPrepPlanes * 64];      // ... and of the int8 one (phase 2)
    for (int i = tid; i < kPrepPlanes * 32; i += 256) s_img4[i] = 0u;
    if (MODE == 1 && tid < 192) reinterpret_cast<unsigned*>(s_lut)[tid] = reinterpret_cast<const unsigned*>(lut)[tid];
    if (tid < (kMaxHues + 1) * 8) s_ok[tid] = pb.ok[tid >> 3][tid & 7];
    if (tid < kPrepPlanes) s_tbl[tid] = pb.tbl[tid];
    const int nblk = (c.P + 255) / 256;                    // pixel ranges per heading
    const int a = blockIdx.x / nblk, blk = blockIdx.x - a * nblk;
    const int px = blk * 256 + tid;
    if (MODE == 1) __syncthreads();                        // (the sensor tables are used in phase 1)
    // ---- phase 1: this thread's pixel
    unsigned H = 0, S = 0, V = 0;
    bool sense_err = false;
    if (px < c.P) {
        if (MODE == 1) {
            if (!sense_pixel(land, g, poses.p[a], s_lut, px / g.sw, px % g.sw, H, S, V)) { sense_err = true; H = S = V = 0; }
        } else if (MODE == 2) {
            synth_hsv(splitmix64((unsigned long long)((long long)a * c.P + px) + (seed + 1ull) * 0x9E3779B97F4A7C15ull), H, S, V, c.synth_full_s);
        } else {
            const unsigned char* r = raw + ((long long)a * c.P + px) * 3;
            H = r[0]; S = r[1]; V = r[2];
        }
    }
    int k_hs = 0, k_bhs = 0, k_bv = 0;
    bool off = false;
    s_raw[tid * 3] = (unsigned char)H; s_raw[tid * 3 + 1] = (unsigned char)S; s_raw[tid * 3 + 2] = (unsigned char)V;
    if (px < c.P) {
        if (!c.generic && c.cw > 0.0) {
            const int nk = c.signed_s ? 2 : c.nhs;
            bool in_set = false;
            for (int k = 0; k < nk; ++k) in_set |= (H == c.hues[k]);
            if (!in_set) k_hs = (int)S;
            else if (c.signed_s && S > 127u) k_hs = (int)S - 127;       // excess over the clamped plane byte
        }
        k_bhs = k_hs;
    }
    for (int pl = 0; pl < c.npl; ++pl) {
        const unsigned av = px < c.P ? plane_byte(c, pl, H, S, V) : 0u;
        s_pl[pl][tid] = (unsigned char)av;
        if (pb.enabled && px < c.P) {
            const int d = abs((int)av - (int)pb.lmin[pl]);
            if (pl < c.nhs) k_bhs += d; else k_bv += d;
        }
    }
    __syncthreads();                                       // s_ok is in LDS; so are the block's bytes
    if (pb.enabled && px < c.P)
        for (int pl = 0; pl < c.npl; ++pl) {
            const unsigned av = s_pl[pl][tid];
            off |= ((s_ok[pl * 8 + (av >> 5)] >> (av & 31)) & 1u) == 0u;
        }
    // the block's sums: wave shuffles, then four values through LDS
    {
        int s0 = k_hs, s1 = k_bhs, s2 = k_bv, s3 = (off ? 1 : 0) | (sense_err ? 2 : 0);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); s3 |= __shfl_xor(s3, o); }
        if ((tid & 63) == 0) { s_red[tid >> 6][0] = s0; s_red[tid >> 6][1] = s1; s_red[tid >> 6][2] = s2; s_red[tid >> 6][3] = s3; }
    }
    // the block's sums, one integer atomic each (after a barrier behind the s_red stores above)
    auto publish = [&]() {
        if (tid == 0 && all) {
            int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int w = 0; w < 4; ++w) { s0 += s_red[w][0]; s1 += s_red[w][1]; s2 += s_red[w][2]; s3 |= s_red[w][3]; }
            const int slot = ((blk & (kAccWays - 1)) * kMaxHeadings + a) * kAccStride;
            if (s0) atomicAdd(&acc->hs[slot], s0);
            if (s1) atomicAdd(&acc->bhs[slot], s1);
            if (s2) atomicAdd(&acc->bv[slot], s2);
            if (s3 & 1) atomicOr(&acc->off, 1u);
            if (s3 & 2) atomicOr(&acc->err, 1ull << (a / A_agent));      // bit = agent of the pass
        }
    };
    // ---- phase 2: out of LDS.  Raw bytes and the byte path's operand dwords first
    const long long rbase = ((long long)a * c.P + (long long)blk * 256) * 3;
    const int npx = c.P - blk * 256 < 256 ? c.P - blk * 256 : 256;      // real pixels of this block
    if (MODE != 0 && all) {
        if ((rbase & 3) == 0 && npx == 256) {
            if (tid < 192) reinterpret_cast<unsigned*>(raw + rbase)[tid] = reinterpret_cast<const unsigned*>(s_raw)[tid];
        } else {
            for (int i = tid; i < npx * 3; i += 256) raw[rbase + i] = s_raw[i];
        }
    }
    if (what & 1)
    for (int i = tid; i < c.npl * 64; i += 256) {
        const int pl = i >> 6, gq = i & 63;                // dword gq of the block: pixels blk * 256 + 4 gq ..
        const int grp = blk * 64 + gq;
        if (grp < c.Q * 4) prep[((long long)pl * c.Q * 4 + grp) * APAD + a] = reinterpret_cast<const unsigned*>(&s_pl[pl][0])[gq];
    }
    // the coefficient images.  The block's 256 pixels are T whole K-steps of a segment with T planes per pixel.  Every thread puts
    // ITS pixel's coefficients where they belong in an LDS copy of the block's entries -- element n = lp T + r of the block (plane
    // r of local pixel lp) is bit beta = n % 32 of dword j = n / 32 % 4 of half n / 128 % 2 of K-step n / 256: the fp4 image keeps it
    // as nibble beta / 4 of the entry of bit position beta % 4 (an LDS atomic OR), the int8 image as byte beta / 8 of the entry of
    // slice beta % 8 (a byte store) -- and the entries then leave as whole 16-byte stores, one per thread.  (Round 3 first built each
    // entry by one thread walking its 32 or 16 elements: 144 threads in long divergent loops, most of the kernel's 28 us at
    // 128x128 x 32 headings.)
    if (pb.enabled && all) {
        const int NKT = pb.NK[0] + pb.NK[1];
        const int pass = a >> 5, al = a & 31;
        const int TT = pb.T[0] + pb.T[1];
        const bool real = tid < npx;
        for (int t = 0; t < TT; ++t) {
            const int seg = t >= pb.T[0] ? 1 : 0;
            const int r = t - (seg ? pb.T[0] : 0), T = pb.T[seg];
            const unsigned tb = s_tbl[t];
            const int n = tid * T + r;
            const int kidx = (seg ? pb.T[0] : 0) + (n >> 8), m = n & 255;
            const int half = m >> 7, j = (m >> 5) & 3, beta = m & 31;
            const int al_ = (int)s_pl[tb & 0xffu][tid] - (int)((tb >> 8) & 0xffu);
            if (pb.fp4) {
                const int wf = (int)(tb >> 24);                            // 0: a copy of a split gap's first plane
                if (real && wf) atomicOr(&s_img4[((kidx * 4 + (beta & 3)) * 2 + half) * 4 + j], (al_ >= wf ? 0xAu : 0x2u) << (4 * (beta >> 2)));
            }
            const int wd = (int)((tb >> 16) & 0xffu);
            const int alpha = al_ < 0 ? 0 : (al_ > wd ? wd : al_);
            reinterpret_cast<unsigned char*>(s_img8)[(((kidx * 8 + (beta & 7)) * 2 + half) * 4 + j) * 4 + (beta >> 3)] = real ? (unsigned char)((wd - 2 * alpha) & 0xff) : (unsigned char)0;
        }
        __syncthreads();
        publish();                                         // (the block's sums: their atomics are under way while the entries are stored)
        const int n4 = pb.fp4 ? 8 : 0;                     // entries per K-step: 4 bit positions x 2 halves (fp4), 8 slices x 2 halves (int8)
        const int per_k = n4 + 16;
        const int total = TT * per_k;
        for (int e = tid; e < total; e += 256) {
            const int kidx = e / per_k, part = e - kidx * per_k;
            const int seg = kidx >= pb.T[0] ? 1 : 0;
            const int kk = kidx - (seg ? pb.T[0] : 0);
            const int ksl = blk * pb.T[seg] + kk;          // K-step within the segment
            if (ksl >= pb.NK[seg]) continue;               // (past the segment's last pixel: stays zero)
            const bool is4 = part < n4;
            const int sub = is4 ? part >> 1 : (part - n4) >> 1, half = part & 1;
            const long long ks = (long long)pass * NKT + (seg ? pb.NK[0] : 0) + ksl;
            if (is4) coef4[(ks * 4 + sub) * 64 + al + 32 * half] = reinterpret_cast<const uint4*>(s_img4)[(kidx * 4 + sub) * 2 + half];
            else coef[(ks * 8 + sub) * 64 + al + 32 * half] = reinterpret_cast<const uint4*>(s_img8)[(kidx * 8 + sub) * 2 + half];
        }
    }
    if (!(pb.enabled && all)) {                            // (with bit planes the sums left in front of the entries' stores)
        __syncthreads();
        publish();
    }
}

// ------------------------------------------------------------------ bit-plane library + int8 MFMA scoring
// |a - b| is not bilinear, but it becomes linear in b once b is known to come from a small level set
// l_0 < l_1 < ... (the reference's sensor quantises V to n_sensor_levels values, NavBySceneFamiliarity.py:176-186, and
// the experiments' saturation is 0 or 127, scripts/run_experiment.py:192).  With the thermometer bits
// B_t = [b >= l_{t+1}] of a library byte and alpha_t = clamp(a - l_t, 0, w_t), w_t = l_{t+1} - l_t, of ANY patch byte a:
//     |a - b| = (l_0 - a)+ + (a - l_max)+ + sum_t alpha_t  +  sum_t B_t * (w_t - 2 alpha_t)
// (gap t contributes its overlap with [min(a,b), max(a,b)]: alpha_t when b is below the gap, w_t - alpha_t when above).
// The first three terms depend on the patch only (a per-heading constant); the last is an exact integer GEMM:
// M = headings, N = views, K = pixels x planes, A = int8 coefficients (gaps wider than 127 are split), B = library BITS.
// So the library is stored as bits (0.75 B/px where the byte planes take 2), and scored on the matrix cores:
//   btiles[g][ks][lane] : uint4 -- view group g = 32 views, K-step ks = 256 K-elements; lane = (view & 31) + 32*half;
//     bit beta of dword j of that lane = K-element ((ks*2 + half)*4 + j)*32 + beta of its segment (HS planes first,
//     then V planes; element n of a segment = plane n % T of pixel n / T; zero beyond the last pixel);
//   coef[ks][s][lane] : uint4 -- A operand of slice s of K-step ks: lane = (heading & 31) + 32*half, byte b of dword j =
//     coefficient of bit beta = s + 8b of library dword j;
//   slice s of a K-step = v_mfma_i32_32x32x32_i8 with B operand (x_j & (0x01010101 << s)), j = 0..3: bytes of value
//     2^s * bit, ONE v_and_b32 per operand dword, the factor 2^s divided out of the accumulator at the end (exact: every
//     term is a multiple of it).  Slices 4..7 use x >> 4, so four accumulators carry the eight slices.
// The integer sums are the ones k_sad_tiles produces, so everything downstream (k_finish / k_combine, tie rule) is shared.
constexpr int kMaxBitPlanes = 16;
static_assert(kMaxBitPlanes == kPrepPlanes, "PrepBits::tbl");
struct BitCfg {
    int T[2];               // planes of the HS segment and of the V segment
    int NK[2];              // K-steps (256 K-elements) of each segment
    int GS;                 // 1-KB rows between consecutive view groups in btiles (>= NK[0] + NK[1], odd)
    int vcode;              // 1: the fp4 form reads the V segment as 3-bit level codes (ctiles, see k_bitpack_code)
    int GSC;                // 256-byte units between view groups in ctiles (>= 4 NK[0] + 3 NK[1])
    int nt;                 // 1: the library rows are streamed with the non-temporal policy (used once per step and larger than the
                            // Infinity Cache); 0: default policy -- a library that fits the 256 MiB cache is re-read from it step after step
    int wacc[2][4];         // fp4 form (sad_ring_fp4): the one gap width of the planes that land on bit b of a nibble, per segment
    int nbp;                // byte planes described below (= LibCfg::npl)
    unsigned char pl[kMaxBitPlanes];   // byte plane of bit plane t (HS planes first, then V)
    unsigned char lo[kMaxBitPlanes];   // lower level of its gap
    unsigned char w[kMaxBitPlanes];    // width of the gap (<= 127)
    unsigned char wfull[kMaxBitPlanes];   // fp4 form: whole distance to the next library level for the FIRST plane of a gap that was
                                          // split to fit int8 coefficients (its copies carry the same bits: they get 0 here and no sign)
    unsigned char lmin[kMaxHues + 1], lmax[kMaxHues + 1];   // per byte plane: smallest / largest library value
};

// presence[pl][v >> 5] bit (v & 31): byte value v occurs in plane pl at a real pixel of a real view
__global__ void __launch_bounds__(256)
k_level_scan(const uint4* __restrict__ tiles, LibCfg c, unsigned* __restrict__ presence) {
    __shared__ unsigned local[(kMaxHues + 1) * 8];
    for (int i = threadIdx.x; i < (kMaxHues + 1) * 8; i += blockDim.x) local[i] = 0;
    __syncthreads();
    const long long total = (c.Fpad / 64) * (long long)c.npl * c.Q * 64;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(t & 63);
        long long r = t >> 6;
        const int q = (int)(r % c.Q); r /= c.Q;
        const int pl = (int)(r % c.npl);
        const long long g = r / c.npl;
        if (g * 64 + lane >= c.F) continue;
        const uint4 v = tiles[g * c.gstride + ((long long)pl * c.Q + q) * 64 + lane];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        unsigned seen[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 16; ++i) {
            if (q * 16 + i >= c.P) break;
            const unsigned b = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
            seen[b >> 5] |= 1u << (b & 31);
        }
        for (int k = 0; k < 8; ++k)
            if (seen[k] & ~local[pl * 8 + k]) atomicOr(&local[pl * 8 + k], seen[k]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (kMaxHues + 1) * 8; i += blockDim.x)
        if (local[i]) atomicOr(&presence[i], local[i]);
}

// K-element n of segment seg -> (bit plane, pixel); false beyond the last pixel
__device__ __forceinline__ bool bit_element(const BitCfg& b, int P, int seg, long long n, int& plane, int& px) {
    const int T = b.T[seg];
    if (T == 0) return false;
    px = (int)(n / T);
    plane = (seg ? b.T[0] : 0) + (int)(n % T);
    return px < P;
}

// byte tiles -> bit tiles.  One thread per (view group of 32, K-step, lane) uint4.
__global__ void __launch_bounds__(256)
k_bitpack(const uint4* __restrict__ tiles, uint4* __restrict__ btiles, LibCfg c, BitCfg b) {
    const int NKT = b.NK[0] + b.NK[1];
    const long long G32 = c.Fpad / 32;
    const long long total = G32 * NKT * 64;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int lane = (int)(t & 63);
    const long long r = t >> 6;
    const int ks = (int)(r % NKT);
    const long long g = r / NKT;
    const long long f = g * 32 + (lane & 31);
    const int half = lane >> 5;
    const int seg = ks >= b.NK[0] ? 1 : 0;
    const int ksl = ks - (seg ? b.NK[0] : 0);
    unsigned w[4] = {0, 0, 0, 0};
    if (f < c.F) {
        const unsigned char* tb = reinterpret_cast<const unsigned char*>(tiles + (f >> 6) * c.gstride + (f & 63));
        for (int j = 0; j < 4; ++j) {
            const long long n0 = (((long long)ksl * 2 + half) * 4 + j) * 32;
            for (int beta = 0; beta < 32; ++beta) {
                int plane, px;
                if (!bit_element(b, c.P, seg, n0 + beta, plane, px)) continue;
                const unsigned byte = tb[((long long)b.pl[plane] * c.Q + (px >> 4)) * 64 * 16 + (px & 15)];
                if (byte >= (unsigned)b.lo[plane] + (unsigned)b.w[plane]) w[j] |= 1u << beta;
            }
        }
    }
    btiles[(g * b.GS + ks) * 64 + lane] = make_uint4(w[0], w[1], w[2], w[3]);
}

// Code tiles: the library of the fp4 form when the value plane has exactly five levels (the reference's default
// n_sensor_levels, NavBySceneFamiliarity.py:66): its four thermometer bits per pixel carry log2(5) bits of information, and
// three stored bits decode back to them with two or three VALU operations per plane --
//     level 0..4  ->  code b2 b1 b0 = 000, 001, 010, 110, 111:   t1 = b0 | b1,  t2 = b1,  t3 = b2,  t4 = b0 & b2.
// The HS rows are the bit tiles' (1 KB: thermometer bits as they are).  A V K-step is THREE code dwords per lane where the
// bit-tile row has four: nibble i of code dword w (w = 0, 1, 2) holds in bits 0..2 the code of the pixel whose thermometer bits
// are nibble i of dword w of the bit-tile row, and in bit 3 bit w of the code of the pixel at nibble i of dword 3.  5 bits per
// pixel where the bit tiles take 6; the K-elements, hence the coefficient image and the sums, are the same.
// A lane's twelve code dwords of a STAGE of four V K-steps (dword d = 3 kk + w of K-step kk) are stored as three full 1-KB rows
// ("chunks": chunk d / 4 holds the lanes' dwords 4 (d / 4) .. + 3, 16 bytes per lane): the kernel moves a stage of a view group
// with three global_load_lds_dwordx4 -- 1 KB per wave-instruction like every other row; rows of 768 B moved by dwordx3 took as long
// as 1-KB rows, sad_lc_fp4 -- and a consumer has a K-step's three dwords in registers after reading whole chunks.
// One thread per (view group of 32, K-step, lane).
__device__ __forceinline__ unsigned level_code(unsigned thermo4) {       // thermometer nibble t4 t3 t2 t1 -> code
    const unsigned level = __popc(thermo4 & 15u);
    return level == 0 ? 0u : level == 1 ? 1u : level == 2 ? 2u : level == 3 ? 6u : 7u;
}
__global__ void __launch_bounds__(256)
k_bitpack_code(const uint4* __restrict__ btiles, unsigned* __restrict__ ctiles, LibCfg c, BitCfg b) {
    const int NKT = b.NK[0] + b.NK[1];                                  // (NK[1] is a whole number of stages of four: build_bit_planes)
    const long long G32 = c.Fpad / 32;
    const long long total = G32 * NKT * 64;
    const long long tt = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (tt >= total) return;
    const int lane = (int)(tt & 63);
    const long long rr = tt >> 6;
    const int ks = (int)(rr % NKT);
    const long long g = rr / NKT;
    const uint4 own = btiles[(g * b.GS + ks) * 64 + lane];
    unsigned* group = ctiles + g * b.GSC * 64;                          // GSC is in 256-byte units = 64 dwords
    if (ks < b.NK[0]) {
        reinterpret_cast<uint4*>(group + (long long)ks * 256)[lane] = own;
    } else {
        const unsigned o[4] = {own.x, own.y, own.z, own.w};
        const int ksl = ks - b.NK[0];
        unsigned* stage = group + (long long)b.NK[0] * 256 + (long long)(ksl >> 2) * 768;      // 3 KB = 768 dwords per stage
        for (int w = 0; w < 3; ++w) {
            unsigned x = 0;
            for (int i = 0; i < 8; ++i) {
                const unsigned code = level_code(o[w] >> (4 * i));
                const unsigned extra = (level_code(o[3] >> (4 * i)) >> w) & 1u;
                x |= (code | (extra << 3)) << (4 * i);
            }
            const int d = 3 * (ksl & 3) + w;
            stage[(d >> 2) * 256 + lane * 4 + (d & 3)] = x;
        }
    }
}

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// The same scoring with BOTH operands streamed through LDS rings by LDS-DMA (global_load_lds, no destination
// registers) RD stages deep, and every vector-memory wait counted by hand:
//   stage = SK K-steps; per stage a wave fetches its share of the coefficient rows (SK*8 rows of 1 KB over the 8 waves)
//   and its own library rows (SK*TILES rows) -- NDMA instructions, all issued in one place at the top of a stage, for
//   stage st + RD - 1, into the ring slot the workgroup finished reading one barrier ago;
//   vmcnt retires in order, so "stage st + 1 has landed" = "at most the NDMA * (RD - 2) younger instructions are
//   still outstanding": one counted s_waitcnt per stage, then a raw s_barrier (every wave's rows are in, every wave is done
//   with the slot about to be refilled).  Nothing ever drains to vmcnt(0) inside a segment, so HBM loads stay RD - 1
//   stages (microseconds) ahead of their use and the accumulators are the only large register block.
// The LDS-DMA instructions are inline assembly: hipcc would wait for vmcnt(0) before every ds_read while one of ITS
// global_load_lds is in flight (it cannot tell the ring slots apart).  Inline assembly is invisible to its waitcnt
// bookkeeping, which is exactly what is wanted here: every wait in the loop is written out below.
template <int N>
__device__ __forceinline__ void wait_vmcnt_le() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_dma_16(const uint4* gsrc, unsigned lds_byte_addr) {     // 64 lanes x 16 B -> 1 KB of LDS
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}
__device__ __forceinline__ void lds_dma_16_nt(const uint4* gsrc, unsigned lds_byte_addr) {  // library rows: used once per step
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// With one chunk the sums of an item never leave the workgroup: FUSE = true turns them into scores right here (k_finish's
// arithmetic, operation for operation), reduces them to the item's summary per heading -- maximum and first view -- and
// lists candidates against the item's own best (the superset rule of k_finish); k_fold behind the kernel folds the
// summaries and decides.  The partial sums then cross HBM not at all, and k_finish / k_combine + k_tail drop out of the step.
struct FuseArgs {
    const int* hsconst;             // per-heading constants of the two sums (k_patch_prep: PrepAcc::bhs, bv, read with acc_sum)
    const int* vconst;
    unsigned long long* bsum;       // [agents][nb][2][A_agent] workgroup summaries
    unsigned long long* ctmp;       // [agents][kTmpCap][2] shared extra-candidate lists
    StepState* st;                  // [agents]
    int A_real;                     // resident headings
    int A_agent;                    // headings per agent
    int nb;                         // summaries per agent (one per workgroup)
    double delta;
};

// Fused finishing of one item (FUSE forms).  The accumulators are transposed there (library bits as the A operand of the
// MFMA): a lane holds ONE heading (a_off + (lane & 31)) and its registers are views, so the reduction over views is an
// in-lane loop.  The score P - sc / 255 falls as sc = whs * shs + wv * sv rises (k_finish's arithmetic, operation for
// operation; division by a positive constant and the subtraction are monotone, roundings included), so the maximum over
// views is taken on sc and only the representatives -- and the few entries near the threshold -- pay for the division:
//   * per heading, the item leaves the key of its best score and a view attaining it (bsum).  Where several views round
//     to that same best score, all of them are within delta of it: whichever is the representative, the others are
//     listed below, and two entries within delta of the step's best send the step to the exact resolver anyway;
//   * every other entry within delta of the item's best (per agent) goes to the shared list (ctmp).  Such an entry is
//     either its lane's smallest (kept in two registers) or within the same margin of it: those -- normally a handful
//     per item, equal sums -- wait in a small LDS queue, so that nothing but a lane's smallest stays in registers across
//     the barriers.  Both are tested on sc against a bound loose by more than every rounding involved, then on the key.
//     Past a full queue they go to the shared list untested (k_fold keeps what reaches the step's threshold).
// `scratch`: LDS, kFuseScratchBytes.  hs / v: the integer sums of this lane (register r = view
// (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of view group t).
constexpr int kFuseQueue = 256;
constexpr int kFuseBlk = 2 * 8 * 32 + 4 * 32 + 2 * kFuseQueue + 2;      // 8-byte words in front of the workgroup's running summary
constexpr int kFuseScratchBytes = (kFuseBlk + 2 * 32) * 8;

template <int TILES, int NW, bool FAST = false, typename HsOf, typename VOf>
__device__ __forceinline__ void
fused_finish(HsOf hs_of, VOf v_of, const long long (&gidx)[TILES], const bool (&live)[TILES],
             unsigned long long* scratch, const LibCfg& c, const FuseArgs& fz, int a_off, int has_hs_sum, long long gq, int lane, int wave,
             int parity, int n_entry_waves = NW, const int* consts = nullptr, unsigned long long* blk = nullptr) {
    static_assert(NW == 8, "scratch layout");
    // Three workgroup barriers per call.  What a call needs cleared on entry is cleared by the call before it (fused_block_begin
    // before the first): abest by wave 0, which alone touches it, and the queue counter of the NEXT call's parity -- the callers
    // alternate `parity`, so a counter is cleared two barriers before anybody adds to it again.
    // Waves that hold no sums (the loader waves of sad_lc_fp4) keep the barriers with fused_finish_idle.  Waves [0, n_entry_waves)
    // hold the entries.
    unsigned long long* sum_sc = scratch;                      // [NW][32] bits of the (non-negative) sc: bit order = value order
    unsigned long long* sum_view = sum_sc + NW * 32;
    unsigned long long* item_view = sum_view + NW * 32;        // [32]
    unsigned long long* abest = item_view + 32;                // [32] per agent of the pass: bits of its smallest sc
    unsigned long long* thr_key = abest + 32;                  // [32] per heading
    unsigned long long* loose_sc = thr_key + 32;               // [32] per heading
    unsigned long long* q_sc = loose_sc + 32;                  // [kFuseQueue] bits of sc
    unsigned long long* q_id = q_sc + kFuseQueue;              // [kFuseQueue] heading of the pass << 40 | view
    unsigned* q_n = reinterpret_cast<unsigned*>(q_id + kFuseQueue) + (parity & 1);      // [2] counters, used alternately
    unsigned* q_n_next = reinterpret_cast<unsigned*>(q_id + kFuseQueue) + ((parity & 1) ^ 1);
    unsigned long long* blk_key = blk ? blk : scratch + kFuseBlk;      // [32] the workgroup's best per heading over its items so far
    unsigned long long* blk_view = blk_key + 32;               // [32] (fused_block_begin / fused_block_end); `blk`: a second heading tile's
    // The lane index is made opaque here: everything below that depends on it is loop-invariant, and the compiler
    // would otherwise compute it once per kernel and hold it in registers through the scoring loop (48 64-bit values
    // per lane: the kernel then spills several hundred bytes per lane and reloads them entry by entry).
    asm volatile("" : "+v"(lane));
    const int half = lane >> 5, n = lane & 31;
    const int agent0 = a_off / fz.A_agent;
    const int a = a_off + n;
    const bool valid = a < fz.A_real;
    const int ac = valid ? a : fz.A_real - 1;
    // the lane's two per-heading constants: loaded here, or once per kernel by the caller (`consts`: one L2 round trip less per item)
    const int hsc = consts ? consts[0] : acc_sum(fz.hsconst, ac), vc = consts ? consts[1] : (fz.vconst ? acc_sum(fz.vconst, ac) : 0);
    // score >= best - delta  =>  sc <= sc_best + 255 delta up to roundings of a few ulp of P: loose by far more
    const double margin = 256. * fz.delta + 256. * (double)c.P * 8.9e-16;
    const double kInf = __longlong_as_double(0x7ff0000000000000ll);
    double bs = kInf;                                          // this lane's smallest sc and its entry
    int bi = 0;                                                // local view index t * 32 + view: ascends with (t, r)
    int nreal[TILES];                                          // real views of each group: entries [0, nreal)
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        const long long left = c.F - gidx[t] * 32;
        nreal[t] = (valid && live[t]) ? (left > 32 ? 32 : (int)left) : 0;
    }
    // The sums fit an int32 (build_bit_planes), so (double)(int) is k_finish's (double)(long long) of the same value.
    auto sc_of = [&](int t, int r) -> double {
        const int shs = hsc + (has_hs_sum ? hs_of(t, r) : 0);
        const int sv = vc + (c.hasv ? v_of(t, r) : 0);
        double sc = c.whs * (double)shs;
        if (c.hasv) sc = sc + c.wv * (double)sv;
        return ((r & 3) + 8 * (r >> 2) + 4 * half) < nreal[t] ? sc : kInf;
    };
    // One pass for the smallest; `near`: some other entry came within the margin of the smallest so far (a superset of
    // "within the margin of the final smallest": the smallest only falls) -- rare, and only then are the entries walked
    // again to queue those within the margin of the final smallest.
    bool near = false;
    // First in fp32 (a quarter of the instructions' cost): sc to ~2.4e-7 relative (four roundings of non-negative terms), its
    // smallest and second smallest value per lane.  Where the second smallest clears the smallest by more than both errors and the
    // margin -- every lane of the wave, nearly always -- the fp32 minimum IS the fp64 one, nothing else lies within the margin of it,
    // and the one fp64 evaluation of that entry gives exactly what the walk below gives.  Otherwise the wave takes the walk.
    // (tools/exp/stamps.py -DDEJAVU_EXP_FIN: the fp64 walk was 3.4-3.9 us of the finishing's 4.6-5.2.)
    bool walked = false;
    if constexpr (FAST) {
        const float whs32 = (float)c.whs, wv32 = (float)c.wv;
        const float kInfF = __int_as_float(0x7f800000);
        const float margin32 = (float)margin * 1.01f + 1e-30f;
        float m32 = kInfF, s32 = kInfF;
        int mi = 0, m_shs = 0, m_sv = 0;
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int vv = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int shs = hsc + (has_hs_sum ? hs_of(t, r) : 0);
                const int sv = vc + (c.hasv ? v_of(t, r) : 0);
                float e = whs32 * (float)shs;
                if (c.hasv) e = __builtin_fmaf(wv32, (float)sv, e);
                e = vv < nreal[t] ? e : kInfF;
                const bool less = e < m32;                     // "<" keeps the first minimum
                s32 = less ? m32 : (e < s32 ? e : s32);
                m_shs = less ? shs : m_shs;
                m_sv = less ? sv : m_sv;
                mi = less ? t * 32 + vv : mi;
                m32 = less ? e : m32;
            }
        const bool unclear = m32 != kInfF && s32 <= m32 * 1.000002f + margin32;
        if (!__any(unclear)) {
            if (m32 != kInfF) {
                double sc = c.whs * (double)m_shs;
                if (c.hasv) sc = sc + c.wv * (double)m_sv;
                bs = sc;
                bi = mi;
            }
            walked = true;
        }
    }
    if (!walked) {
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double sc = sc_of(t, r);
            const bool less = sc < bs;                         // "<" keeps the first minimum
            near |= less ? (bs <= sc + margin) : (sc <= bs + margin && sc != kInf);
            if (less) { bs = sc; bi = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half; }
        }
    }
    if (__any(near)) {
        const double lim = bs + margin;
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int vv = (r & 3) + 8 * (r >> 2) + 4 * half;
                const double sc = sc_of(t, r);
                if (sc <= lim && sc != kInf && t * 32 + vv != bi) {      // (a lane without entries has lim = +inf)
                    const unsigned pos = atomicAdd(q_n, 1u);
                    if (pos < (unsigned)kFuseQueue) {
                        q_sc[pos] = (unsigned long long)__double_as_longlong(sc);
                        q_id[pos] = ((unsigned long long)n << 40) | (unsigned long long)(gidx[t] * 32 + vv);
                    } else {
                        // the queue is full (a library of duplicates): straight to the shared list, untested -- k_fold
                        // keeps what reaches the step's threshold
                        const int agent = a / fz.A_agent, kk = a - agent * fz.A_agent;
                        const unsigned gpos = __hip_atomic_fetch_add(&fz.st[agent].ntmp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (gpos < (unsigned)kTmpCap) {
                            unsigned long long* ct = fz.ctmp + (long long)agent * kTmpCap * 2;
                            ct[2 * gpos] = ((unsigned long long)kk << 40) | (unsigned long long)(gidx[t] * 32 + vv);
                            ct[2 * gpos + 1] = ordered_key((double)c.P - sc / 255.);
                        }
                    }
                }
            }
    }
#ifdef DEJAVU_EXP_FIN
    if (threadIdx.x == 0 && gq == 0) g_dv_stamps[(blockIdx.x & 255) * 8 + 6] = __builtin_amdgcn_s_memrealtime();
#endif
    unsigned long long own_v = ~0ull;                          // the view of the lane's smallest
#pragma unroll
    for (int t = 0; t < TILES; ++t)
        if ((bi >> 5) == t) own_v = (unsigned long long)(gidx[t] * 32 + (bi & 31));
    const double own_s = bs;
    {
        const double os = __shfl_xor(bs, 32);
        const unsigned long long ov = __shfl_xor(own_v, 32);
        unsigned long long bv = own_v;
        if (os < bs || (os == bs && ov < bv)) { bs = os; bv = ov; }
        if (half == 0) { sum_sc[wave * 32 + n] = (unsigned long long)__double_as_longlong(bs); sum_view[wave * 32 + n] = bv; }
    }
    __syncthreads();
    if (threadIdx.x == 32) *q_n_next = 0;                      // (the next call's queue counter: nobody reads or adds to it during this call)
    const unsigned long long kNone = 0x7ff0000000000000ull;   // +inf: no entry (sc >= 0: bit order = value order)
    if (threadIdx.x < 32) {                                    // thread n: the item's summary of heading a_off + n
        unsigned long long is = kNone, iv = ~0ull;
        if (valid) {
            for (int i = 0; i < n_entry_waves; ++i) {
                const unsigned long long k = sum_sc[i * 32 + n], w = sum_view[i * 32 + n];
                if (k < is || (k == is && w < iv)) { is = k; iv = w; }
            }
            const int agent = a / fz.A_agent, kk = a - agent * fz.A_agent;
            const bool any = is != kNone;
            if (any) {
                atomicMin(&abest[agent - agent0], is);
                // The workgroup keeps ONE summary per heading over all its items (256 summaries per step instead of one per
                // item): the item's representative against the workgroup's so far.  The loser is listed when it is within
                // delta of the winner -- a superset of "within delta of the step's best", which is no smaller than the winner.
                const unsigned long long ck = ordered_key((double)c.P - __longlong_as_double((long long)is) / 255.), cv = iv;
                const unsigned long long bk = blk_key[n], bv = blk_view[n];
                if (bk == 0) {
                    blk_key[n] = ck; blk_view[n] = cv;
                } else {
                    const bool cur_wins = ck > bk || (ck == bk && cv < bv);
                    const unsigned long long wk = cur_wins ? ck : bk, lk = cur_wins ? bk : ck, lv = cur_wins ? bv : cv;
                    if (key_to_double(lk) >= key_to_double(wk) - fz.delta) {
                        const unsigned pos = __hip_atomic_fetch_add(&fz.st[agent].ntmp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (pos < (unsigned)kTmpCap) {
                            unsigned long long* ct = fz.ctmp + (long long)agent * kTmpCap * 2;
                            ct[2 * pos] = ((unsigned long long)kk << 40) | lv;
                            ct[2 * pos + 1] = lk;
                        }
                    }
                    if (cur_wins) { blk_key[n] = ck; blk_view[n] = cv; }
                }
            } else {
                iv = ~0ull;
            }
            (void)gq;
        }
        item_view[n] = iv;
        // Same 32 threads, same wave: its LDS operations complete in order, so every lane's atomicMin above has been applied when
        // the loads below execute -- no barrier between the two halves of this section.
        unsigned long long tk = ~0ull, ls = 0ull;              // nothing passes
        if (valid) {
            const unsigned long long best = __hip_atomic_load(&abest[a / fz.A_agent - agent0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (best != ~0ull) {
                const double scb = __longlong_as_double((long long)best);
                tk = ordered_key(((double)c.P - scb / 255.) - fz.delta);
                ls = (unsigned long long)__double_as_longlong(scb + margin);
            }
        }
        thr_key[n] = tk;
        loose_sc[n] = ls;
        __hip_atomic_store(&abest[n], ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // clear for the next call (this wave alone touches it)
    }
    __syncthreads();
#ifdef DEJAVU_EXP_FIN
    if (threadIdx.x == 0 && gq == 0) g_dv_stamps[(blockIdx.x & 255) * 8 + 7] = __builtin_amdgcn_s_memrealtime();
#endif
    auto list = [&](int m, double sc, unsigned long long f) {  // heading a_off + m of the pass, an entry within the loose bound
        const unsigned long long k = ordered_key((double)c.P - sc / 255.);
        if (k >= thr_key[m] && f != item_view[m]) {
            const int am = a_off + m, agent = am / fz.A_agent, kk = am - agent * fz.A_agent;
            const unsigned pos = __hip_atomic_fetch_add(&fz.st[agent].ntmp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pos < (unsigned)kTmpCap) {
                unsigned long long* ct = fz.ctmp + (long long)agent * kTmpCap * 2;
                ct[2 * pos] = ((unsigned long long)kk << 40) | f;
                ct[2 * pos + 1] = k;
            }
        }
    };
    if (own_s <= __longlong_as_double((long long)loose_sc[n])) list(n, own_s, own_v);      // (+inf = no entry; the bound is finite)
    const unsigned qn = *q_n;
    for (unsigned i = threadIdx.x; i < (qn < (unsigned)kFuseQueue ? qn : (unsigned)kFuseQueue); i += blockDim.x) {
        const int m = (int)(q_id[i] >> 40);
        const double sc = __longlong_as_double((long long)q_sc[i]);
        if (sc <= __longlong_as_double((long long)loose_sc[m])) list(m, sc, q_id[i] & ((1ull << 40) - 1));
    }
    __syncthreads();               // the scratch is reused by the next item
}

// What a wave without sums does while the others finish an item: fused_finish's three workgroup barriers, raw (nothing waits for the
// LDS-DMA a loader wave has in flight for the next item).
__device__ __forceinline__ void fused_finish_idle() {
#pragma unroll
    for (int i = 0; i < 3; ++i) __builtin_amdgcn_s_barrier();
}

// The workgroup's running summary (fused_finish): cleared before its first item, written after its last --
// bsum[agent][workgroup][2][A_agent], nb = gridDim.x summaries per agent.
__device__ __forceinline__ void fused_block_begin(unsigned long long* scratch, bool first = true) {
    unsigned long long* blk_key = scratch + kFuseBlk;
    if (threadIdx.x < 32) { blk_key[threadIdx.x] = 0; blk_key[32 + threadIdx.x] = ~0ull; }
    if (first) {                                               // what fused_finish expects cleared on entry
        unsigned long long* abest = scratch + 2 * 8 * 32 + 32;
        if (threadIdx.x < 32) abest[threadIdx.x] = ~0ull;
        if (threadIdx.x == 32) scratch[2 * 8 * 32 + 4 * 32 + 2 * kFuseQueue] = 0ull;      // both queue counters
    }
    __syncthreads();
}
__device__ __forceinline__ void fused_block_end(const unsigned long long* scratch, const FuseArgs& fz, const LibCfg& c, int a_off) {
    const unsigned long long* blk_key = scratch + kFuseBlk;
    __syncthreads();
    const int n = threadIdx.x, a = a_off + n;
    if (n < 32 && a < fz.A_real) {
        const int agent = a / fz.A_agent, kk = a - agent * fz.A_agent;
        unsigned long long* bsm = fz.bsum + ((long long)agent * gridDim.x + blockIdx.x) * 2 * fz.A_agent;
        __hip_atomic_store(&bsm[kk], blk_key[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&bsm[fz.A_agent + kk], blk_key[n] ? blk_key[32 + n] : ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    (void)c;                                      // k_fold, launched behind this kernel, folds the summaries and decides
}

template <int SK, int TILES, int RD, bool FUSE>
__device__ __forceinline__ void
sad_ring_i8(const uint4* __restrict__ btiles, const uint4* __restrict__ coef, int* __restrict__ part, const LibCfg& c, const BitCfg& b,
            int nchunk, int apad_total, int a_off, int has_hs_sum, const FuseArgs& fz, int n_gq) {
    extern __shared__ uint4 lds_ring[];           // [RD][ coefficient rows SK*8 | library rows 8 waves * SK * TILES ][64], then FUSE scratch
    constexpr int NW = 8;
    constexpr int COEF_ROWS = SK * 8;             // per stage
    constexpr int LIB_ROWS = NW * SK * TILES;
    constexpr int SLOT16 = (COEF_ROWS + LIB_ROWS) * 64;      // uint4 per ring slot
    constexpr int CPW = COEF_ROWS / NW;           // coefficient rows each wave fetches per stage
    constexpr int NDMA = CPW + SK * TILES;        // LDS-DMA instructions per wave and stage
    static_assert(COEF_ROWS % NW == 0 && RD >= 2 && NDMA * (RD - 1) < 64, "ring shape");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long G32 = c.Fpad / 32;
    const long long GQ = n_gq;                    // view-group ranges the library is cut into (item_groups)
    const long long n_items = GQ * nchunk;
    const int rows = (apad_total - a_off) < 32 ? (apad_total - a_off) : 32;
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)lds_ring;
    if constexpr (FUSE) fused_block_begin(reinterpret_cast<unsigned long long*>(lds_ring + RD * SLOT16));
    int nfin = 0;                                 // fused_finish calls so far (their parity)
    (void)nfin;
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ch = (int)(item / GQ);
        const long long gq = item - (long long)ch * GQ;
        const long long g0 = (gq * G32) / GQ, g1 = ((gq + 1) * G32) / GQ;     // at most VW groups (launch_mfma)
        const uint4* lib[TILES];
        long long gidx[TILES];
        bool live[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            long long g = g0 + wave * TILES + t;
            live[t] = g < g1;
            if (!live[t]) g = g0;                 // a slot without a group streams nothing: it re-reads one row (L2), see issue_stage
            gidx[t] = g;
            lib[t] = btiles + (g * b.GS) * 64 + lane;
        }
        int parkr[TILES][16];
        (void)parkr;
#pragma unroll 1
        for (int seg = 0; seg < 2; ++seg) {
            if (seg == 0 && !has_hs_sum) continue;
            if (seg == 1 && !c.hasv) continue;
            const int kbase = seg ? b.NK[0] : 0;
            const int k0 = kbase + (int)(((long long)ch * b.NK[seg]) / nchunk);
            const int k1 = kbase + (int)(((long long)(ch + 1) * b.NK[seg]) / nchunk);
            const int nst = (k1 - k0 + SK - 1) / SK;
            v16i_t acc[TILES][4];
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][s][r] = 0;
            if (nst > 0) {
                auto kclamp = [&](int k) { return k < k1 ? k : k1 - 1; };
                // all of a stage's LDS-DMA of this wave: CPW coefficient rows, then its SK*TILES library rows
                auto issue_stage = [&](int st) {
                    const int kb = k0 + st * SK;
                    const unsigned slot = lds_base + (unsigned)((st % RD) * SLOT16) * 16u;
#pragma unroll
                    for (int i = 0; i < CPW; ++i) {
                        const int row = wave * CPW + i;                       // (K-step, slice) row of the stage
                        long long src = ((long long)kb * 8 + row) * 64 + lane;
                        const long long lim = (long long)k1 * 512;
                        if (src >= lim) src = lim - 64 + lane;                // past the chunk: any valid row (masked below)
                        lds_dma_16(coef + src, __builtin_amdgcn_readfirstlane(slot + (unsigned)(row * 64) * 16u));
                    }
#pragma unroll
                    for (int k = 0; k < SK; ++k)
#pragma unroll
                        for (int t = 0; t < TILES; ++t) {
                            const int row = COEF_ROWS + (wave * SK + k) * TILES + t;
                            const unsigned dst = __builtin_amdgcn_readfirstlane(slot + (unsigned)(row * 64) * 16u);
                            if (live[t]) lds_dma_16_nt(lib[t] + (long long)kclamp(kb + k) * 64, dst);
                            else lds_dma_16(coef + (long long)k0 * 512 + lane, dst);     // a coefficient row, hot in L2: keeps this wave's vmcnt arithmetic; never used
                        }
                };
#pragma unroll
                for (int r = 0; r < RD - 1; ++r) issue_stage(r);              // stages 0 .. RD-2 (clamped rows past the end)
                for (int st = 0; st < nst; ++st) {
                    // stage st has landed once at most the (RD - 2) younger stages' instructions are outstanding
                    wait_vmcnt_le<NDMA * (RD - 2)>();
                    __builtin_amdgcn_s_barrier();
                    issue_stage(st + RD - 1);                                  // into the slot everybody finished with a barrier ago
                    const int kb = k0 + st * SK;
                    const uint4* cbuf = lds_ring + (st % RD) * SLOT16 + lane;
                    const uint4* lbuf = cbuf + (COEF_ROWS + wave * SK * TILES) * 64;
                    v4i_t a[2][4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) { const uint4 w = cbuf[s * 64]; a[0][s] = v4i_t{(int)w.x, (int)w.y, (int)w.z, (int)w.w}; }
                    uint4 xl[2][TILES];                                        // library bits of the current / next K-step
#pragma unroll
                    for (int t = 0; t < TILES; ++t) xl[0][t] = lbuf[t * 64];
#pragma unroll
                    for (int hs = 0; hs < 2 * SK; ++hs) {
                        const int k = hs >> 1;
                        if (hs + 1 < 2 * SK) {
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const uint4 w = cbuf[((hs + 1) * 4 + s) * 64];
                                a[(hs + 1) & 1][s] = v4i_t{(int)w.x, (int)w.y, (int)w.z, (int)w.w};
                            }
                            if ((hs & 1) == 0 && k + 1 < SK) {
#pragma unroll
                                for (int t = 0; t < TILES; ++t) xl[(k + 1) & 1][t] = lbuf[((k + 1) * TILES + t) * 64];
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const bool on = kb + k < k1;
#pragma unroll
                        for (int t = 0; t < TILES; ++t) {
                            const uint4 x = xl[k & 1][t];
                            const uint4 src = (hs & 1) ? make_uint4(x.x >> 4, x.y >> 4, x.z >> 4, x.w >> 4) : x;
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const unsigned m = on ? (0x01010101u << s) : 0u;
                                const v4i_t bo = v4i_t{(int)(src.x & m), (int)(src.y & m), (int)(src.z & m), (int)(src.w & m)};
                                if constexpr (FUSE) acc[t][s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(bo, a[hs & 1][s], acc[t][s], 0, 0, 0);   // views x headings
                                else acc[t][s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[hs & 1][s], bo, acc[t][s], 0, 0, 0);
                            }
                        }
                    }
                }
                wait_vmcnt_le<0>();               // the clamped fetches past the end of the chunk
                __builtin_amdgcn_s_barrier();     // nobody still reads a slot the next segment / item refills
            }
            const int type_row = seg ? has_hs_sum : 0;
            const int nsum = has_hs_sum + c.hasv;
            int tot[TILES][16];
#pragma unroll
            for (int t = 0; t < TILES; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    tot[t][r] = acc[t][0][r] + (acc[t][1][r] >> 1) + (acc[t][2][r] >> 2) + (acc[t][3][r] >> 3);
            if constexpr (!FUSE) {
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    if (live[t]) {
                        int* dst = part + ((long long)(ch * nsum + type_row) * apad_total + a_off) * c.Fpad + gidx[t] * 32 + (lane & 31);
                        int half = lane >> 5;                       // opaque: the row addresses are not hoisted out of the item loop (sad_lc_fp4's store_sums)
                        asm volatile("" : "+v"(half));
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (m < rows) dst[(long long)m * c.Fpad] = tot[t][r];
                        }
                    }
                }
            } else {
                // ---- fused finishing (one chunk): the saturation sums wait in registers for the value sums
                if (seg == 0 && c.hasv) {
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
#pragma unroll
                        for (int r = 0; r < 16; ++r) parkr[t][r] = tot[t][r];
                } else {
                    unsigned long long* scratch = reinterpret_cast<unsigned long long*>(lds_ring + RD * SLOT16);
                    auto of_tot = [&](int t, int r) -> int { return tot[t][r]; };
                    auto of_park = [&](int t, int r) -> int { return parkr[t][r]; };
                    if (seg == 0) fused_finish<TILES, NW, false>(of_tot, of_tot, gidx, live, scratch, c, fz, a_off, has_hs_sum, gq, lane, wave, nfin);
                    else fused_finish<TILES, NW, false>(of_park, of_tot, gidx, live, scratch, c, fz, a_off, has_hs_sum, gq, lane, wave, nfin);
                    ++nfin;
                }
            }
        }
    }
    if constexpr (FUSE) fused_block_end(reinterpret_cast<const unsigned long long*>(lds_ring + RD * SLOT16), fz, c, a_off);
}

// The fp4 form of the same kernel.  When every patch byte sits on a level or outside the library's range (the
// reference's patches come through the quantiser the library came through, NavBySceneFamiliarity.py:176-186), every
// coefficient w_t - 2 alpha_t is +w_t or -w_t; and when the planes that land on bit position b of a nibble all have one
// gap width w_b (K-element n = plane n % T sits on bit n % 4: always so for 1, 2 or 4 planes per pixel, else when the
// widths agree; a gap wider than 127 was split for the int8 form into planes with the same bits: its first plane
// stands for the whole gap here and the copies get coefficient 0), the sum over the K-elements on bit b is
//     sum B (w - 2 alpha) = w_b * sum B * (+-1):
// the signs are exact in fp4 (E2M1: +-1.0), the library bits too (bit 0/1/2 of a nibble ARE the E2M1 values 0.5/1/2, bit 3
// comes down by a shift), and v_mfma_f32_32x32x64_f8f6f4 multiplies 64 K-elements in the time the int8 form takes for 32
// (measured: 21.2 vs 18.3 ns per instruction and SIMD, tools/exp/mfma_fp4.hip).  The SAME bit tiles are the B operand:
// K-element <-> (dword j, bit 4i + b) pairs with nibble i of coefficient image b of that K-step (k_patch_prep), and any
// pairing works as long as both operands use it.  Sums of +-{0.5, 1, 2} stay exact in the fp32 accumulators (below 2^24
// in halves); the four accumulators (one per bit position) are multiplied by their widths as integers at the end: the int32 sums are
// the int8 form's, bit for bit.  With the coefficient image half the size, the kernel is left to the HBM stream.
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));

// LDS operand reads of the fp4 loop, written out: hipcc sinks ordinary LDS loads to their first use when registers are
// tight (254 of 256 here), which put a full LDS round trip in front of every second MFMA (0.94 ms at 500 000 views whatever
// the ring depth, stage length or HBM bytes).  The reads below stay where they are issued -- one K-step ahead of their use --
// and lds_tie after the counted wait keeps everything that uses them below it.  What the compiler must NOT do is copy or spill
// such a register between the read and its wait (it believes the value is there): whole-register outputs into locals are
// not copied, and tools/kernel_resources.py shows the scratch size (0 bytes in the loops; three ds_read_b32 into the
// elements of one vector WERE copied out early in one build, which is why code rows are read as b128 too).
template <int OFF>
__device__ __forceinline__ void lds_read16(v4u_t& dst, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_tie(v4u_t& a) { asm volatile("" : "+v"(a)); }
template <int K> struct IntC { static constexpr int value = K; };
template <int N, int I = 0, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) { f(IntC<I>{}); static_for<N, I + 1>(f); }
}

// One segment (HS or V) of one item in the fp4 form: the ring loop, then the segment's integer sums in tot.  Library rows are
// the bit tiles' (thermometer bits, 1 KB per K-step; the code tiles are sad_lc_fp4's).
// [k0, k1): K-steps of the coefficient image (absolute); lib[t]: this lane's place in the library row of K-step k0.
// w[b]: gap width of the planes on bit b.
// kflush in (k0, k1), a whole number of stages past k0: the range covers BOTH segments (HS rows then V rows, consecutive
// in the bit tiles and in the coefficient image) -- at K-step kflush the accumulators leave the HS sums in totf (widths wf)
// and start over, so the ring streams through the boundary and an item has one pipeline fill, not two.
template <int SK, int TILES, int RD, bool FUSE>
__device__ __forceinline__ void
fp4_segment(const unsigned char* const (&lib)[TILES], const bool (&live)[TILES], const uint4* __restrict__ coef4, int k0, int k1, const int (&w)[4],
            int lane, int wave, int (&tot)[TILES][16], int kflush, const int (&wf)[4], int (&totf)[TILES][16]) {
    extern __shared__ uint4 lds_ring[];
    constexpr int NW = 8;
    constexpr int ROWB = 1024, LROWB = 1024;      // bytes of a library row in HBM and in LDS
    constexpr int COEF_ROWS = SK * 4;
    constexpr int LIB_ROWS = NW * SK * TILES;
    constexpr int SLOTB = COEF_ROWS * 1024 + LIB_ROWS * LROWB;
    constexpr int CPW = COEF_ROWS / NW;
    constexpr int NDMA = CPW + SK * TILES;
    static_assert(COEF_ROWS % NW == 0 && RD >= 2 && NDMA * (RD - 1) < 64, "ring shape");
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)lds_ring;
    const int nst = (k1 - k0 + SK - 1) / SK;
    v16f_t acc[TILES][4];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][s][r] = 0.f;
    if (nst > 0) {
        // DMA instruction d of stage st: CPW coefficient rows of this wave, then its SK * TILES library rows
        auto issue_one = [&](int st, int d) {
            const int kb = k0 + st * SK;
            const unsigned slot = lds_base + (unsigned)((st % RD) * SLOTB);
            if (d < CPW) {
                const int row = wave * CPW + d;                       // (K-step, bit position) row of the stage
                long long src = ((long long)kb * 4 + row) * 64 + lane;
                const long long lim = (long long)k1 * 256;
                if (src >= lim) src = lim - 64 + lane;                // past the chunk: any valid row (masked below)
                lds_dma_16(coef4 + src, __builtin_amdgcn_readfirstlane(slot + (unsigned)row * 1024u));
            } else {
                const int k = (d - CPW) / TILES, t = (d - CPW) % TILES;
                const int row = (wave * SK + k) * TILES + t;
                int lr = st * SK + k;
                lr = lr < k1 - k0 ? lr : k1 - k0 - 1;                 // clamped: masked below
                const unsigned dst = __builtin_amdgcn_readfirstlane(slot + (unsigned)(COEF_ROWS * 1024 + row * LROWB));
                if (!live[t]) lds_dma_16(coef4 + (long long)k0 * 256 + lane, dst);       // a slot without a group: a coefficient row, hot in L2
                else lds_dma_16_nt(reinterpret_cast<const uint4*>(lib[t] + (long long)lr * ROWB), dst);
            }
        };
        auto issue_stage = [&](int st) {
#pragma unroll
            for (int d = 0; d < NDMA; ++d) issue_one(st, d);
        };
        static_assert(SK % 2 == 0, "operand buffers alternate by K-step across stages");
        // Operand registers of two K-steps: (stage, k) lives in buffer k & 1 (SK is even), read from LDS one K-step ahead.
        // XSTAGE (one view group per wave) -- ALSO across stages: the wait + barrier that publish stage st + 1 sit in front
        // of the LAST K-step of stage st, whose operands are in registers by then, so the first operands of stage st + 1
        // are on their way while that K-step runs and the matrix pipes do not idle behind every barrier for an LDS round
        // trip of all eight waves (100 000 views x 64x64 x 64 headings: 146 -> 133 us).  With two view groups per wave the
        // registers do not reach: the compiler then reloads an LDS address from scratch in the loop, and the vmcnt(0) it puts
        // behind that reload drains the ring every stage (500 000 views x 128x128: 0.975 instead of 0.95 ms) -- so there the
        // barrier stays at the top of the stage.
        constexpr bool XSTAGE = TILES == 1;
        constexpr int READS = 4 + TILES;              // LDS reads per K-step: 4 coefficient rows, TILES library rows
        v4u_t a[2][4], xl[2][TILES];
        auto fetch = [&](int st, auto kc) {
            constexpr int k = decltype(kc)::value;
            const unsigned sad = lds_base + (unsigned)((st % RD) * SLOTB);
            const unsigned cad = sad + (unsigned)lane * 16u;                                              // coefficient row r: + 1024 r
            const unsigned lad = sad + (unsigned)(COEF_ROWS * 1024 + wave * SK * TILES * LROWB) + (unsigned)lane * 16u;
            static_for<4>([&](auto sc) { constexpr int s_ = decltype(sc)::value; lds_read16<(k * 4 + s_) * 1024>(a[k & 1][s_], cad); });
            static_for<TILES>([&](auto tc) {
                constexpr int t_ = decltype(tc)::value;
                lds_read16<(k * TILES + t_) * LROWB>(xl[k & 1][t_], lad);
            });
        };
#pragma unroll
        for (int r = 0; r < RD - 1; ++r) issue_stage(r);
        if constexpr (XSTAGE) {
            wait_vmcnt_le<NDMA * (RD - 2)>();         // stage 0 has landed once at most the younger stages' DMA is outstanding
            __builtin_amdgcn_s_barrier();
            DV_STAMP(1);
            fetch(0, IntC<0>{});
        }
        for (int st = 0; st < nst; ++st) {
            if constexpr (!XSTAGE) {
                wait_vmcnt_le<NDMA * (RD - 2)>();     // stage st has landed ...
                __builtin_amdgcn_s_barrier();         // ... for everybody, and everybody is done with the slot refilled below
                fetch(st, IntC<0>{});
            }
            if (k0 + st * SK == kflush && st > 0) {                      // segment boundary (uniform, once per item)
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        totf[t][r] = __mul24(wf[0], (int)(2.f * acc[t][0][r])) + __mul24(wf[1], (int)acc[t][1][r]) + __mul24(wf[2], (int)(0.5f * acc[t][2][r])) +
                                     __mul24(wf[3], (int)acc[t][3][r]);
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[t][s][r] = 0.f;
            }
            // The DMA of stage st + RD - 1 goes into the slot everybody finished with at the previous barrier, one instruction
            // after every few MFMAs (a burst of 8 x NDMA wave-instructions behind a barrier queues on the CU's one
            // vector-memory path while every matrix pipe waits); XSTAGE: all of it before the last K-step.
            int dma_next = 0;
            constexpr int MFMAS = (XSTAGE ? SK - 1 : SK) * TILES * 4;
            constexpr int EVERY = MFMAS / NDMA > 0 ? MFMAS / NDMA : 1;
            int mfma_count = 0;
            const int kb = k0 + st * SK;
            auto mfma = [&](int t, int s, const v4u_t& av, unsigned b0, unsigned b1, unsigned b2, unsigned b3) {
                const v8i_t bo = v8i_t{(int)b0, (int)b1, (int)b2, (int)b3, 0, 0, 0, 0};
                const v8i_t ao = v8i_t{(int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0};
                if constexpr (FUSE) acc[t][s] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bo, ao, acc[t][s], 4, 4, 0, 0, 0, 0);   // views x headings
                else acc[t][s] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ao, bo, acc[t][s], 4, 4, 0, 0, 0, 0);
                if (++mfma_count % EVERY == 0 && dma_next < NDMA) issue_one(st + RD - 1, dma_next++);
            };
            static_for<SK>([&](auto kc) {
                constexpr int k = decltype(kc)::value;
                if constexpr (k + 1 < SK) {
                    fetch(st, IntC<k + 1>{});             // one K-step ahead
                    lds_wait<READS>();                    // all but the newest READS reads have landed: K-step k's
                } else {
                    lds_wait<0>();                        // this wave holds everything it will use of slot st % RD
                    if constexpr (XSTAGE) {
#pragma unroll
                        for (int d = 0; d < NDMA; ++d)
                            if (d >= dma_next) issue_one(st + RD - 1, d);     // (none left when the MFMAs divide evenly)
                        dma_next = NDMA;
                        if (st + 1 < nst) {
                            wait_vmcnt_le<NDMA * (RD - 2)>(); // stage st + 1 has landed (this wave's rows) ...
                            __builtin_amdgcn_s_barrier();     // ... for everybody, and everybody is done reading slot st % RD
                            fetch(st + 1, IntC<0>{});
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) lds_tie(a[k & 1][s]);
#pragma unroll
                for (int t = 0; t < TILES; ++t) lds_tie(xl[k & 1][t]);
                const bool on = kb + k < k1;
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    const unsigned x[4] = {xl[k & 1][t].x, xl[k & 1][t].y, xl[k & 1][t].z, xl[k & 1][t].w};
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        // bit s of every nibble as an E2M1 value: 0.5 / 1 / 2 in place, bit 3 shifted down to the 1.0 position
                        const unsigned m = on ? (s < 3 ? (0x11111111u << s) : 0x22222222u) : 0u;
                        const int sh = s < 3 ? 0 : 2;
                        mfma(t, s, a[k & 1][s], (x[0] >> sh) & m, (x[1] >> sh) & m, (x[2] >> sh) & m, (x[3] >> sh) & m);
                    }
                }
            });
            if constexpr (!XSTAGE) {
#pragma unroll
                for (int d = 0; d < NDMA; ++d)
                    if (d >= dma_next) issue_one(st + RD - 1, d);         // (none left when the MFMAs divide evenly)
            }
        }
        wait_vmcnt_le<0>();
        __builtin_amdgcn_s_barrier();
        DV_STAMP(2);
    }
    // bits stood for 0.5 / 1 / 2 / 1: signed counts 2 acc0, acc1, acc2 / 2, acc3 -- integers -- each times the gap width of the
    // planes on that bit position (widths < 256, |counts| < 2^23)
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            tot[t][r] = __mul24(w[0], (int)(2.f * acc[t][0][r])) + __mul24(w[1], (int)acc[t][1][r]) + __mul24(w[2], (int)(0.5f * acc[t][2][r])) +
                        __mul24(w[3], (int)acc[t][3][r]);
}

// LDS the rings of fp4_segment take (the FUSE scratch sits behind the larger of a kernel's two).
constexpr int fp4_ring_bytes(int SK, int TILES, int RD) { return RD * (SK * 4 * 1024 + 8 * SK * TILES * 1024); }

template <int SK, int TILES, int RD, bool FUSE>
__device__ __forceinline__ void
sad_ring_fp4(const uint4* __restrict__ btiles, const uint4* __restrict__ coef4, int* __restrict__ part, const LibCfg& c, const BitCfg& b,
             int nchunk, int apad_total, int a_off, int has_hs_sum, const FuseArgs& fz, int n_gq) {
    extern __shared__ uint4 lds_ring[];           // the rings of fp4_segment, then the FUSE scratch
    constexpr int NW = 8;
    constexpr int RING = fp4_ring_bytes(SK, TILES, RD);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long G32 = c.Fpad / 32;
    const long long GQ = n_gq;                    // view-group ranges the library is cut into (item_groups)
    const long long n_items = GQ * nchunk;
    const int rows = (apad_total - a_off) < 32 ? (apad_total - a_off) : 32;
    const long long gbytes = (long long)b.GS * 1024;                    // between view groups
    DV_STAMP(0);
    if constexpr (FUSE) fused_block_begin(reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(lds_ring) + RING));
    int nfin = 0;                                 // fused_finish calls so far (their parity)
    (void)nfin;
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ch = (int)(item / GQ);
        const long long gq = item - (long long)ch * GQ;
        const long long g0 = (gq * G32) / GQ, g1 = ((gq + 1) * G32) / GQ;     // at most VW groups (launch_mfma)
        const unsigned char* grp[TILES];
        long long gidx[TILES];
        bool live[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            long long g = g0 + wave * TILES + t;
            live[t] = g < g1;
            if (!live[t]) g = g0;                 // a slot without a group streams nothing (fp4_segment: one row, re-read)
            gidx[t] = g;
            grp[t] = reinterpret_cast<const unsigned char*>(btiles) + g * gbytes;
        }
        int tot_hs[TILES][16], tot_v[TILES][16];
        (void)tot_hs;
        (void)tot_v;
        // one chunk, both sums, HS K-steps a whole number of stages: ONE ring loop for the item
        const bool merged = nchunk == 1 && has_hs_sum && c.hasv && b.NK[0] % SK == 0 && b.NK[0] > 0 && b.NK[1] > 0;
        if (merged) {
            const unsigned char* lib[TILES];
#pragma unroll
            for (int t = 0; t < TILES; ++t) lib[t] = grp[t] + lane * 16;
            fp4_segment<SK, TILES, RD, FUSE>(lib, live, coef4, 0, b.NK[0] + b.NK[1], b.wacc[1], lane, wave, tot_v, b.NK[0], b.wacc[0], tot_hs);
        }
        if (!merged && has_hs_sum) {
            const int k0 = (int)(((long long)ch * b.NK[0]) / nchunk), k1 = (int)(((long long)(ch + 1) * b.NK[0]) / nchunk);
            const unsigned char* lib[TILES];
#pragma unroll
            for (int t = 0; t < TILES; ++t) lib[t] = grp[t] + (long long)k0 * 1024 + lane * 16;
            fp4_segment<SK, TILES, RD, FUSE>(lib, live, coef4, k0, k1, b.wacc[0], lane, wave, tot_hs, -1, b.wacc[0], tot_hs);
        }
        if (!merged && c.hasv) {
            const int r0 = (int)(((long long)ch * b.NK[1]) / nchunk), r1 = (int)(((long long)(ch + 1) * b.NK[1]) / nchunk);
            const unsigned char* lib[TILES];
#pragma unroll
            for (int t = 0; t < TILES; ++t) lib[t] = grp[t] + (long long)(b.NK[0] + r0) * 1024 + lane * 16;
            fp4_segment<SK, TILES, RD, FUSE>(lib, live, coef4, b.NK[0] + r0, b.NK[0] + r1, b.wacc[1], lane, wave, tot_v, -1, b.wacc[1], tot_v);
        }
        if constexpr (!FUSE) {
            const int nsum = has_hs_sum + c.hasv;
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                if (live[t]) {
#pragma unroll
                    for (int seg = 0; seg < 2; ++seg) {
                        if (seg == 0 ? !has_hs_sum : !c.hasv) continue;
                        const int type_row = seg ? has_hs_sum : 0;
                        int* dst = part + ((long long)(ch * nsum + type_row) * apad_total + a_off) * c.Fpad + gidx[t] * 32 + (lane & 31);
                        int half = lane >> 5;                       // opaque, as above
                        asm volatile("" : "+v"(half));
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (m < rows) dst[(long long)m * c.Fpad] = seg ? tot_v[t][r] : tot_hs[t][r];
                        }
                    }
                }
            }
        } else {
            unsigned long long* scratch = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(lds_ring) + RING);
            auto of_hs = [&](int t, int r) -> int { return tot_hs[t][r]; };
            auto of_v = [&](int t, int r) -> int { return tot_v[t][r]; };
            if (item == blockIdx.x) DV_STAMP(3);
            fused_finish<TILES, NW, false>(of_hs, of_v, gidx, live, scratch, c, fz, a_off, has_hs_sum, gq, lane, wave, nfin++);
            if (item == blockIdx.x) DV_STAMP(4);
        }
    }
    if constexpr (FUSE) fused_block_end(reinterpret_cast<const unsigned long long*>(reinterpret_cast<unsigned char*>(lds_ring) + RING), fz, c, a_off);
    DV_STAMP(5);
}

// ------------------------------------------------------------------ fp4 body with loader and consumer waves
// The ring loop above makes every wave do everything: issue its LDS-DMA (an M0 dance and ~100 cycles of issue each), wait on
// vmcnt, read operands from LDS, mask, multiply.  tools/exp/stamps.py shows what that costs: a workgroup takes ~300 ns per
// K-step whatever the bytes it moves (50 000 views x 64x64: 28.7 us of ring loop for 8 view groups, 30.3 for 6-7; deeper rings
// and the 3-bit code rows change nothing), 2.4x the matrix pipe's time -- the loop is bound by its own instruction stream.
// Here the roles are split.  Waves 0-3, one per SIMD, are CONSUMERS: two view groups each (so every coefficient row read from
// LDS serves two tiles and the workgroup reads each row four times, not eight), no vector-memory instruction in the loop, only
// ds_read_b128 one K-step ahead, the masks and the MFMAs.  Waves 4-7, their partners on the SIMDs, are LOADERS: all LDS-DMA
// of a stage (SK K-steps: 4 coefficient rows + 8 library rows each) split four ways, counted vmcnt waits, and ONE s_barrier
// per stage that publishes stage st and frees the slot of stage st - 1 (the consumers take it in front of their last K-step
// of stage st - 1, whose operands are in registers by then, and fetch the first operands of stage st behind it).
// The loaders number stages through ALL items of the workgroup: while the consumers finish an item (fused_finish, in which
// the loaders only keep the barriers), the first RD - 1 stages of the next item are already in flight -- no pipeline fill
// per item.  Past the workgroup's last item the loaders issue re-reads of a coefficient row (hot in L2) so that the counted
// waits stay uniform.  One range of K-steps [0, NKT) with the accumulators flushed at the segment boundary (kflush), so one
// chunk only; K chunks and the int8 form keep the loop above.
template <int SK, int RD>
constexpr int lc_ring_bytes() { return RD * SK * 12 * 1024; }

template <int SK, int RD, bool FUSE, bool CODE, int HT>
__device__ __forceinline__ void
sad_lc_fp4(const uint4* __restrict__ ftiles, const uint4* __restrict__ coef4, int* __restrict__ part, const LibCfg& c, const BitCfg& b,
           int apad_total, int a_off, int has_hs_sum, const FuseArgs& fz, int n_gq) {
    // HT = 1: a consumer multiplies TWO view groups by the 32 headings at a_off (8 view groups per item);
    // HT = 2: ONE view group by the 64 headings at a_off .. a_off + 63, two heading tiles whose coefficient images lie one
    //         pass apart (4 view groups per item): the library crosses HBM once for 64 headings, and the masked library
    //         operands are made once for both tiles.
    // Either way a K-step is 12 rows of 1 KB in the ring (4 HT coefficient rows + 8 / HT library rows) and 8 MFMAs per consumer.
    extern __shared__ uint4 lds_ring[];
    constexpr int TL = 2 / HT;                                          // view groups per consumer
    constexpr int NW = 8, NC = 4, NL = 4;
    constexpr int KCOEF = 4 * HT, KLIB = NC * TL;                       // rows per K-step
    constexpr int COEF_ROWS = SK * KCOEF, ROWS = SK * (KCOEF + KLIB);
    constexpr int SLOTB = ROWS * 1024;
    constexpr int PER = ROWS / NL;                                      // LDS-DMA instructions per loader wave and stage
    constexpr int RING = RD * SLOTB;
    constexpr int NU = SK * HT;                                         // pipeline units per stage: (K-step, heading tile)
    constexpr int PERC = SK * HT + 3 * TL;                              // ... of a stage of code rows: three 1-KB chunks per view group
    static_assert((HT == 1 || HT == 2) && PER == 3 * SK && PER * (RD - 1) < 64 && SK % 2 == 0 && RD >= 3, "ring shape");
    static_assert(!CODE || (SK == 4 && RD == 3), "code rows come in stages of four K-steps; the counted wait below knows one younger stage");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave >= NC;
    const long long G32 = c.Fpad / 32, GQ = n_gq;
    const int NKT = b.NK[0] + b.NK[1];
    const int nst = (NKT + SK - 1) / SK;
    // code tiles (k_bitpack_code): a V stage of a view group is three 1-KB chunks (a lane's twelve 3-bit-code dwords of the
    // stage's four K-steps), chunk c in the ring row where K-step c's library row would be, decoded in registers by the consumers.
    // (Rows of 768 B per K-step moved by global_load_lds_dwordx3 took LONGER than the thermometer rows they replace: 500 000 views
    // x 128x128, first item's loop 128 us against 114 -- tools/exp/stamps.py.)
    constexpr bool codev = CODE;                                        // == (b.vcode != 0): the host picks the instantiation
    const int NK0 = b.NK[0];
    const long long gbytes = codev ? (long long)b.GSC * 256 : (long long)b.GS * 1024;      // between view groups
    const long long pass16 = (long long)NKT * 256;                      // uint4 between the coefficient images of two heading tiles
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)lds_ring;
    unsigned long long* scratch0 = reinterpret_cast<unsigned long long*>(reinterpret_cast<unsigned char*>(lds_ring) + RING);
    const unsigned char* lib_bytes = reinterpret_cast<const unsigned char*>(ftiles);
    DV_STAMP(0);
    if constexpr (FUSE) {
#pragma unroll
        for (int h = 0; h < HT; ++h) fused_block_begin(scratch0 + h * 64, h == 0);      // (running summaries of heading tile h: 64 words behind the first tile's)
    }
    const long long n_mine = GQ > (long long)blockIdx.x ? (GQ - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;       // items of this workgroup

    // ---- loader state: the stage it issues next, (li, lst) = (item of this workgroup, stage), into ring slot lslot
    const int lw = wave - NC;
    long long li = 0;
    int lst = 0, lslot = 0;
    const unsigned char* lp[TL];                                        // this lane's place in row 0 of the loader's view group(s)
    const unsigned char* lpv[TL];                                       // and in their first V chunk when those are code rows
    bool llive[TL];
#pragma unroll
    for (int t = 0; t < TL; ++t) { lp[t] = lib_bytes; lpv[t] = lib_bytes; llive[t] = false; }
    auto loader_item = [&]() {                                          // (re)aim at item li
#pragma unroll
        for (int t = 0; t < TL; ++t) llive[t] = false;
        if (li < n_mine) {
            const long long item = blockIdx.x + li * gridDim.x;
            const long long g0 = (item * G32) / GQ, g1 = ((item + 1) * G32) / GQ;
#pragma unroll
            for (int t = 0; t < TL; ++t) {
                const long long g = g0 + lw + 4 * t;                    // slot lw + 4 t of the item's 4 TL: consumer (lw + 4 t) / TL, its group (lw + 4 t) % TL
                llive[t] = g < g1;
                lp[t] = lib_bytes + (llive[t] ? g : g0) * gbytes + lane * 16;
                lpv[t] = lib_bytes + (llive[t] ? g : g0) * gbytes + (long long)NK0 * 1024 + lane * 16;
            }
        }
    };
    bool young_code = false;                                            // the stage issued last is one of code rows (PERC instructions, not PER)
    const bool nt_rows = b.nt != 0;
    auto issue_stage = [&]() {
        const unsigned slot = lds_base + (unsigned)lslot * (unsigned)SLOTB;
        const int kb = lst * SK;
        const uint4* hot = coef4 + lw * 64 + lane;                      // re-read where there is nothing to fetch (hot in L2)
        const bool code_stage = codev && kb >= NK0;
#pragma unroll
        for (int i = 0; i < SK * HT; ++i) {                             // coefficient row (K-step kb + kk, heading tile h, bit position lw)
#ifdef DEJAVU_EXP_SKIP
            if (DEJAVU_EXP_SKIP & 1) continue;
#endif
            const int kk = i / HT, h = i % HT;
            int k = kb + kk;
            k = k < NKT ? k : NKT - 1;
            const unsigned dst = __builtin_amdgcn_readfirstlane(slot + (unsigned)((kk * KCOEF + h * 4 + lw) * 1024));
            lds_dma_16(li < n_mine ? coef4 + h * pass16 + ((long long)k * 4 + lw) * 64 + lane : hot, dst);
        }
        if (code_stage) {
#pragma unroll
            for (int i = 0; i < 3 * TL; ++i) {                          // chunk cc of the stage, slot lw + 4 t
#ifdef DEJAVU_EXP_SKIP
                if (DEJAVU_EXP_SKIP & 2) continue;
#endif
                const int cc = i / TL, t = i % TL;
                const unsigned dst = __builtin_amdgcn_readfirstlane(slot + (unsigned)((COEF_ROWS + cc * KLIB + lw + 4 * t) * 1024));
                if (!llive[t]) lds_dma_16(hot, dst);
                else if (nt_rows) lds_dma_16_nt(reinterpret_cast<const uint4*>(lpv[t] + (long long)((kb - NK0) / SK) * 3072 + cc * 1024), dst);
                else lds_dma_16(reinterpret_cast<const uint4*>(lpv[t] + (long long)((kb - NK0) / SK) * 3072 + cc * 1024), dst);
            }
        } else {
#pragma unroll
            for (int i = 0; i < SK * TL; ++i) {                         // library row (K-step kb + kk, slot lw + 4 t)
#ifdef DEJAVU_EXP_SKIP
                if (DEJAVU_EXP_SKIP & 2) continue;
#endif
                const int kk = i / TL, t = i % TL;
                int k = kb + kk;
                k = k < NKT ? k : NKT - 1;
                const unsigned dst = __builtin_amdgcn_readfirstlane(slot + (unsigned)((COEF_ROWS + kk * KLIB + lw + 4 * t) * 1024));
                if (!llive[t]) lds_dma_16(hot, dst);
                else if (nt_rows) lds_dma_16_nt(reinterpret_cast<const uint4*>(lp[t] + (long long)k * 1024), dst);
                else lds_dma_16(reinterpret_cast<const uint4*>(lp[t] + (long long)k * 1024), dst);
            }
        }
        young_code = code_stage;
        lslot = lslot + 1 == RD ? 0 : lslot + 1;
        if (++lst == nst) { lst = 0; ++li; loader_item(); }
    };
    // The two kinds of waves run SEPARATE item loops that meet at the barriers: what a loader keeps across an item (its pointers, its
    // place in the stream) is then not live through the consumers' loop and the other way round -- with ONE loop around both bodies
    // the allocator spilled loader values into the stage loop as soon as the finishing grew, and the s_waitcnt vmcnt(0) behind every
    // reload drained the LDS-DMA stream each stage (two heading tiles: 1.14 -> 1.96 ms per ensemble step).
    if (loader) {
        loader_item();
#pragma unroll
        for (int r = 0; r < RD - 1; ++r) issue_stage();
        for (long long j = 0; j < n_mine; ++j) {
            for (int st = 0; st < nst; ++st) {
                // this wave's rows of stage (j, st) have landed once only the younger stages' instructions are outstanding ...
#ifdef DEJAVU_EXP_SKIP          // (timing experiments of tools/exp/stamps.py: bit 0 leaves out the coefficient rows, bit 1 the library rows)
                constexpr int XC = (DEJAVU_EXP_SKIP & 1) ? 0 : SK * HT, XL = (DEJAVU_EXP_SKIP & 2) ? 0 : SK * TL, XLC = (DEJAVU_EXP_SKIP & 2) ? 0 : 3 * TL;
                if (CODE && young_code) wait_vmcnt_le<XC + XLC>();
                else wait_vmcnt_le<(XC + XL) * (RD - 2)>();
#else
                if (CODE && young_code) wait_vmcnt_le<PERC>();
                else wait_vmcnt_le<PER * (RD - 2)>();
#endif
                __builtin_amdgcn_s_barrier();                           // ... everybody's have; nobody still reads the slot before it
                issue_stage();                                          // (may belong to the next item: its pipeline fill)
            }
            if constexpr (FUSE) {                                       // the barriers of the consumers' finishing (fused_finish: three per call)
#pragma unroll
                for (int h = 0; h < HT; ++h)
                    if (HT == 1 || a_off + 32 * h < fz.A_real) fused_finish_idle();
            }
        }
        wait_vmcnt_le<0>();                                             // the re-reads past the last item
    } else {

    int cslot = 0;                                                      // ring slot of the consumers' current stage
    int nfin = 0;                                                       // fused_finish calls so far (HT = 2: their parity)
    (void)nfin;
    int hconst[HT][2];                                                  // this lane's heading constants (fused_finish), per heading tile
#pragma unroll
    for (int h = 0; h < HT; ++h) { hconst[h][0] = 0; hconst[h][1] = 0; }
    if (FUSE) {
#pragma unroll
        for (int h = 0; h < HT; ++h) {
            const int a = a_off + 32 * h + (lane & 31), ac = a < fz.A_real ? a : fz.A_real - 1;
            hconst[h][0] = acc_sum(fz.hsconst, ac);
            hconst[h][1] = fz.vconst ? acc_sum(fz.vconst, ac) : 0;
        }
    }
    for (long long j = 0; j < n_mine; ++j) {
        const long long item = blockIdx.x + j * gridDim.x;
        const long long g0 = (item * G32) / GQ, g1 = ((item + 1) * G32) / GQ;
        long long gidx[TL];
        bool live[TL];
#pragma unroll
        for (int t = 0; t < TL; ++t) {
            const long long g = g0 + wave * TL + t;
            live[t] = g < g1;
            gidx[t] = live[t] ? g : g0;
        }
        // integer sums of the item: [u][r], u = view group t (HT = 1) or heading tile h (HT = 2)
        int tot_hs[2][16], tot_v[2][16];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) { tot_hs[u][r] = 0; tot_v[u][r] = 0; }
        // Unfused passes (k_finish behind the kernel: want_scene steps, the mixed layout) store a segment's sums as soon as its
        // accumulators are flushed -- the saturation sums at the segment boundary, not at the item's end: kept live through the V
        // stages beside the 128 accumulators, the operand buffers and the stores' addresses they cost the loop 36-76 bytes of
        // scratch per lane (tools/kernel_resources.py; tests/test_host_logic.py now refuses any)
        auto store_sums = [&](const int (&tot)[2][16], int seg) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int t = HT == 1 ? u : 0, h = HT == 1 ? 0 : u;
                const int rows = (apad_total - a_off - 32 * h) < 32 ? (apad_total - a_off - 32 * h) : 32;
                if (!live[t]) continue;
                const int type_row = seg ? has_hs_sum : 0;
                int* dst = part + ((long long)type_row * apad_total + a_off + 32 * h) * c.Fpad + gidx[t] * 32 + (lane & 31);
                // (the lane's row offset is made opaque here: the sixteen 64-bit row addresses are loop invariants, and hoisted out of the
                // item loop they sat in 32 registers through the stage loop -- the spills tools/kernel_resources.py showed)
                int half = lane >> 5;
                asm volatile("" : "+v"(half));
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < rows) dst[(long long)m * c.Fpad] = tot[u][r];
                }
            }
        };
        {
            v16f_t acc[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[u][s][r] = 0.f;
            // operand registers of two units: unit q = (K-step q / HT, heading tile q % HT) lives in buffer q & 1; the library rows of
            // a K-step in xl[k & 1] (read with the K-step's first unit)
            // Code rows: chunk 0 of a stage is read where K-step 0's row would be (so the read ahead across a stage boundary is the
            // same whatever the next stage holds), chunk 1 with K-step 1, chunk 2 with K-step 2 into xc2 (chunk 0 is still in use
            // then: K-step 1 takes its last dword), nothing with K-step 3.
            v4u_t a[2][4], xl[2][TL], xc2[TL];
            auto fetch = [&](int slot_i, auto qc, auto codeC) {
                constexpr int q = decltype(qc)::value;
                constexpr int k = q / HT, h = q % HT;
                constexpr bool code_rows = decltype(codeC)::value != 0;
                const unsigned sad = lds_base + (unsigned)slot_i * (unsigned)SLOTB + (unsigned)lane * 16u;
#ifdef DEJAVU_EXP_NOLDS
                if (q > 0) return;
#endif
                static_for<4>([&](auto sc) { constexpr int s_ = decltype(sc)::value; lds_read16<(k * KCOEF + h * 4 + s_) * 1024>(a[q & 1][s_], sad); });
                if constexpr (h == 0 && (!code_rows || k < 3)) {
                    const unsigned lad = sad + (unsigned)(COEF_ROWS * 1024 + wave * TL * 1024);
                    static_for<TL>([&](auto tc) {
                        constexpr int t_ = decltype(tc)::value;
                        if constexpr (code_rows && k == 2) lds_read16<(k * KLIB + t_) * 1024>(xc2[t_], lad);
                        else lds_read16<(k * KLIB + t_) * 1024>(xl[k & 1][t_], lad);
                    });
                }
            };
            // bits stood for 0.5 / 1 / 2 / 1 (code rows: 0.5): signed counts 2 acc0, acc1, acc2 / 2, acc3 (code rows: 2 acc3) -- integers
            auto flush = [&](int (&dst)[2][16], const int (&wd)[4], bool code) {
                const float f3 = code ? 2.f : 1.f;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dst[u][r] = __mul24(wd[0], (int)(2.f * acc[u][0][r])) + __mul24(wd[1], (int)acc[u][1][r]) + __mul24(wd[2], (int)(0.5f * acc[u][2][r])) +
                                    __mul24(wd[3], (int)(f3 * acc[u][3][r]));
            };
            __builtin_amdgcn_s_barrier();                               // stage (j, 0) is in LDS
            if (j == 0) DV_STAMP(1);
            fetch(cslot, IntC<0>{}, IntC<0>{});
            // stages [s0, s1) of one kind: thermometer rows, or (codeC) the 3-bit code rows of the V segment
            auto run_stages = [&](int s0, int s1, auto codeC) {
                constexpr bool code_stage = decltype(codeC)::value;
                for (int st = s0; st < s1; ++st) {
                    const int kb = st * SK;
                    const int nslot = cslot + 1 == RD ? 0 : cslot + 1;
                    unsigned bo[4][4];                                  // HT = 2: the masked library operands of the K-step, made once for both heading tiles
                    static_for<NU>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        constexpr int k = q / HT, h = q % HT;
                        if constexpr (q + 1 < NU) {
                            fetch(cslot, IntC<q + 1>{}, codeC);         // one unit ahead
                            constexpr int k1 = (q + 1) / HT, h1 = (q + 1) % HT;
                            lds_wait<4 + ((h1 == 0 && (!code_stage || k1 < 3)) ? TL : 0)>();      // all but the newest unit's reads have landed
                        } else {
                            lds_wait<0>();                              // everything this wave will use of the slot is in registers
                            if (st + 1 < nst) {
                                __builtin_amdgcn_s_barrier();           // stage st + 1 is in LDS; the loaders may refill slot st - 1 ... and,
                                fetch(nslot, IntC<0>{}, IntC<0>{});     //   one barrier later, this one
                            }
                        }
#pragma unroll
                        for (int s = 0; s < 4; ++s) lds_tie(a[q & 1][s]);
                        if constexpr (h == 0) {
#pragma unroll
                            for (int t = 0; t < TL; ++t) {
                                if constexpr (!code_stage || k < 2) lds_tie(xl[k & 1][t]);
                                else if constexpr (k == 2) lds_tie(xc2[t]);
                            }
                        }
                        const bool on = kb + k < NKT;
                        auto mfma = [&](int u, int s, unsigned b0, unsigned b1, unsigned b2, unsigned b3) {
                            const v8i_t bv = v8i_t{(int)b0, (int)b1, (int)b2, (int)b3, 0, 0, 0, 0};
                            const v4u_t& av = a[q & 1][s];
                            const v8i_t ao = v8i_t{(int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0};
                            if constexpr (FUSE) acc[u][s] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bv, ao, acc[u][s], 4, 4, 0, 0, 0, 0);   // views x headings
                            else acc[u][s] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ao, bv, acc[u][s], 4, 4, 0, 0, 0, 0);
                        };
                        // operand of bit position s from the library dwords x (thermometer rows), or of thermometer plane s (code rows)
                        auto operand = [&](const unsigned (&x)[4], int s, unsigned (&o)[4]) {
#ifdef DEJAVU_EXP_NOMASK
                            if (true) {
#pragma unroll
                                for (int d = 0; d < 4; ++d) o[d] = x[d];
                                return;
                            }
#endif
                            if constexpr (!code_stage) {
                                // bit s of every nibble as an E2M1 value: 0.5 / 1 / 2 in place, bit 3 shifted down to the 1.0 position
                                const unsigned m = on ? (s < 3 ? (0x11111111u << s) : 0x22222222u) : 0u;
                                const int sh = s < 3 ? 0 : 2;
#pragma unroll
                                for (int d = 0; d < 4; ++d) o[d] = (x[d] >> sh) & m;
                            } else {
                                // E2M1 operands of the four thermometer planes: t1, t4 as 0.5 (bit 0), t2 as 1 (bit 1), t3 as 2 (bit 2)
                                const unsigned m = on ? 0x11111111u : 0u;
#pragma unroll
                                for (int d = 0; d < 4; ++d)
                                    o[d] = s == 0 ? (x[d] | (x[d] >> 1)) & m : s == 1 ? x[d] & (m << 1) : s == 2 ? x[d] & (m << 2) : x[d] & (x[d] >> 2) & m;
                            }
                        };
                        auto load_x = [&](int t, unsigned (&x)[4]) {
                            if constexpr (!code_stage) {
                                x[0] = xl[k & 1][t].x; x[1] = xl[k & 1][t].y; x[2] = xl[k & 1][t].z; x[3] = xl[k & 1][t].w;
                            } else {
                                // the K-step's three code dwords: dwords 3 k .. 3 k + 2 of the stage's twelve (chunks xl[0], xl[1], xc2)
                                if constexpr (k == 0) { x[0] = xl[0][t].x; x[1] = xl[0][t].y; x[2] = xl[0][t].z; }
                                else if constexpr (k == 1) { x[0] = xl[0][t].w; x[1] = xl[1][t].x; x[2] = xl[1][t].y; }
                                else if constexpr (k == 2) { x[0] = xl[1][t].z; x[1] = xl[1][t].w; x[2] = xc2[t].x; }
                                else { x[0] = xc2[t].y; x[1] = xc2[t].z; x[2] = xc2[t].w; }
                                // the fourth dword's pixels: bit 3 of the three code dwords' nibbles
                                x[3] = ((x[0] >> 3) & 0x11111111u) | ((x[1] >> 2) & 0x22222222u) | ((x[2] >> 1) & 0x44444444u);
                            }
                        };
                        if constexpr (HT == 1) {
#pragma unroll
                            for (int t = 0; t < TL; ++t) {
                                unsigned x[4];
                                load_x(t, x);
#pragma unroll
                                for (int s = 0; s < 4; ++s) {
                                    unsigned o[4];
                                    operand(x, s, o);
                                    mfma(t, s, o[0], o[1], o[2], o[3]);
                                }
                            }
                        } else {
                            if constexpr (h == 0) {
                                unsigned x[4];
                                load_x(0, x);
#pragma unroll
                                for (int s = 0; s < 4; ++s) operand(x, s, bo[s]);
                            }
#pragma unroll
                            for (int s = 0; s < 4; ++s) mfma(h, s, bo[s][0], bo[s][1], bo[s][2], bo[s][3]);
                        }
                    });
                    cslot = nslot;
                }
            };
            // HS stages (when the library has that segment), then the V stages; the accumulators change hands at the boundary
            const int nst0 = has_hs_sum ? (NK0 / SK < nst ? NK0 / SK : nst) : 0;       // (NK0 is a whole number of stages)
            const int hs_end = (has_hs_sum && !(c.hasv && b.NK[1] > 0)) ? nst : nst0; // (no V K-steps: all of them)
            run_stages(0, hs_end, IntC<0>{});
            if (hs_end > 0 && hs_end < nst) {
                flush(tot_hs, b.wacc[0], false);
                if constexpr (!FUSE) store_sums(tot_hs, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[u][s][r] = 0.f;
            }
            if constexpr (CODE) run_stages(hs_end, nst, IntC<1>{});
            else run_stages(hs_end, nst, IntC<0>{});
            if (hs_end < nst) flush(tot_v, b.wacc[1], codev);           // what the accumulators hold at the end: the V segment's sums,
            else flush(tot_hs, b.wacc[0], false);                       // unless there are no V K-steps
            if constexpr (!FUSE) {
                // (a sum without K-steps is stored as the zeros it was initialised to: the mixed layout's saturation rows, a constant value plane)
                if (has_hs_sum && !(hs_end > 0 && hs_end < nst)) store_sums(tot_hs, 0);
                if (c.hasv) store_sums(tot_v, 1);
            }
        }
        if (j == 0) DV_STAMP(2);
        if constexpr (FUSE) {
            if (j == 0) DV_STAMP(3);
            if constexpr (HT == 1) {
                auto of_hs = [&](int t, int r) -> int { return tot_hs[t][r]; };
                auto of_v = [&](int t, int r) -> int { return tot_v[t][r]; };
                fused_finish<TL, NW, true>(of_hs, of_v, gidx, live, scratch0, c, fz, a_off, has_hs_sum, j, lane, wave, (int)(j & 1), NC, hconst[0]);
            } else {
#pragma unroll
                for (int h = 0; h < HT; ++h) {
                    auto of_hs = [&](int, int r) -> int { return tot_hs[h][r]; };
                    auto of_v = [&](int, int r) -> int { return tot_v[h][r]; };
                    if (a_off + 32 * h < fz.A_real)                     // (uniform: a heading tile without headings has nothing to finish)
                        fused_finish<TL, NW, true>(of_hs, of_v, gidx, live, scratch0, c, fz, a_off + 32 * h, has_hs_sum, j, lane, wave, nfin++ & 1, NC,
                                             hconst[h], scratch0 + kFuseBlk + h * 64);
                }
            }
            if (j == 0) DV_STAMP(4);
        }
    }
    }                                                                   // (consumers)
    if constexpr (FUSE) {
#pragma unroll
        for (int h = 0; h < HT; ++h)
            if (a_off + 32 * h < fz.A_real) fused_block_end(scratch0 + h * 64, fz, c, a_off + 32 * h);
    }
    DV_STAMP(5);
}

// ------------------------------------------------------------------ loader / consumer body, two view groups x two heading tiles
// sad_lc_fp4 with two heading tiles (HT = 2: the ensemble passes of 64 headings) gives a consumer ONE view group: per K-step it reads
// eight coefficient rows and one library row from LDS for eight MFMAs, and the CU's LDS moves 4 x 9 KB of reads + 12 KB of LDS-DMA =
// 1.5 KB per MFMA where 128 B/cycle x 32 cycles / 4 SIMDs = 1 KB would keep the matrix pipe fed: the loop runs at ~626 cycles per
// K-step against 256 of MFMA time, bound by LDS bandwidth (neither the matrix pipe nor HBM: 3.5 TB/s).  Here a consumer takes TWO
// view groups and both tiles: the eight coefficient rows serve sixteen MFMAs, (8 + 2) KB of reads + 16 KB of DMA per 64 MFMAs =
// 0.875 KB per MFMA.  Sixteen accumulators of 16 registers would not fit a wave (2 groups x 2 tiles x 4 bit positions), so the
// bit positions of one gap width SHARE an accumulator: every position's library bit is moved to the 1.0 bit of its nibble (a shift
// and a mask; sad_lc_fp4 leaves bits 0 / 1 / 2 where they stand for 0.5 / 1 / 2 and weights the four accumulators at the end), and
// the sums of +-1 of positions 1, 2, 3 -- whose widths must agree (the host checks: 63, 64, 64, 64 of the five sensor levels do) --
// accumulate in one register block, position 0 in the other: 8 blocks = 128 registers.  The saturation segment's counts (one width:
// the host checks) wait for the value segment in LDS as int16 (|count| <= K-elements of the segment <= 32767: the host checks),
// 8 KB per consumer, so that nothing but the accumulators and the operands lives through the stage loop.
// Same ring protocol as sad_lc_fp4 (stage = SK K-steps of 16 rows: 8 coefficient + 8 library; loaders 4 SK instructions per stage),
// fused finishing per heading tile with both view groups (fused_finish<2>).  FUSE forms only, thermometer rows only.
template <int SK, int RD>
constexpr int lc22_ring_bytes() { return RD * SK * 16 * 1024; }
constexpr int kLc22ParkBytes = 4 * 64 * 64 * 2;                         // 4 consumers x [2 groups][2 tiles][16] counts x 64 lanes x int16

template <int SK, int RD>
__device__ __forceinline__ void
sad_lc22_fp4(const uint4* __restrict__ ftiles, const uint4* __restrict__ coef4, const LibCfg& c, const BitCfg& b, int apad_total, int a_off,
             int has_hs_sum, const FuseArgs& fz, int n_gq) {
    extern __shared__ uint4 lds_ring[];
    constexpr int HT = 2, TL = 2;
    constexpr int NW = 8, NC = 4, NL = 4;
    constexpr int KCOEF = 4 * HT, KLIB = NC * TL;
    constexpr int COEF_ROWS = SK * KCOEF, ROWS = SK * (KCOEF + KLIB);
    constexpr int SLOTB = ROWS * 1024;
    constexpr int PER = ROWS / NL;                                      // LDS-DMA instructions per loader wave and stage
    constexpr int RING = RD * SLOTB;
    constexpr int NU = SK * HT;                                         // pipeline units per stage: (K-step, heading tile)
    static_assert(PER == SK * (HT + TL) && PER * (RD - 1) < 64 && RD >= 3, "ring shape");
    static_assert(RING + kLc22ParkBytes + kFuseScratchBytes + 512 <= 160 * 1024, "LDS");
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool loader = wave >= NC;
    const long long G32 = c.Fpad / 32, GQ = n_gq;
    const int NKT = b.NK[0] + b.NK[1];
    const int nst = (NKT + SK - 1) / SK;
    const int NK0 = b.NK[0];
    const long long gbytes = (long long)b.GS * 1024;                    // between view groups
    const long long pass16 = (long long)NKT * 256;                      // uint4 between the coefficient images of two heading tiles
    const unsigned lds_base = (unsigned)(unsigned long long)(lds_ptr_t)lds_ring;
    unsigned char* lds_bytes = reinterpret_cast<unsigned char*>(lds_ring);
    short* park = reinterpret_cast<short*>(lds_bytes + RING) + (wave & 3) * (64 * 64);     // this consumer's saturation counts
    unsigned long long* scratch0 = reinterpret_cast<unsigned long long*>(lds_bytes + RING + kLc22ParkBytes);
    const unsigned char* lib_bytes = reinterpret_cast<const unsigned char*>(ftiles);
    DV_STAMP(0);
#pragma unroll
    for (int h = 0; h < HT; ++h) fused_block_begin(scratch0 + h * 64, h == 0);
    const long long n_mine = GQ > (long long)blockIdx.x ? (GQ - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;       // items of this workgroup

    if (loader) {
        // ---- loaders: the stage they issue next, (li, lst) = (item of this workgroup, stage), into ring slot lslot
        const int lw = wave - NC;
        long long li = 0;
        int lst = 0, lslot = 0;
        const unsigned char* lp[TL];
        bool llive[TL];
#pragma unroll
        for (int t = 0; t < TL; ++t) { lp[t] = lib_bytes; llive[t] = false; }
        auto loader_item = [&]() {
#pragma unroll
            for (int t = 0; t < TL; ++t) llive[t] = false;
            if (li < n_mine) {
                const long long item = blockIdx.x + li * gridDim.x;
                const long long g0 = (item * G32) / GQ, g1 = ((item + 1) * G32) / GQ;
#pragma unroll
                for (int t = 0; t < TL; ++t) {
                    const long long g = g0 + lw + 4 * t;                // slot lw + 4 t of the item's 8: consumer (lw + 4 t) / 2, its group (lw + 4 t) % 2
                    llive[t] = g < g1;
                    lp[t] = lib_bytes + (llive[t] ? g : g0) * gbytes + lane * 16;
                }
            }
        };
        const bool nt_rows = b.nt != 0;
        auto issue_stage = [&]() {
            const unsigned slot = lds_base + (unsigned)lslot * (unsigned)SLOTB;
            const int kb = lst * SK;
            const uint4* hot = coef4 + lw * 64 + lane;                  // re-read where there is nothing to fetch (hot in L2)
#pragma unroll
            for (int i = 0; i < SK * HT; ++i) {                         // coefficient row (K-step kb + kk, heading tile h, bit position lw)
                const int kk = i / HT, h = i % HT;
                int k = kb + kk;
                k = k < NKT ? k : NKT - 1;
                const unsigned dst = __builtin_amdgcn_readfirstlane(slot + (unsigned)((kk * KCOEF + h * 4 + lw) * 1024));
#ifdef DEJAVU_EXP22             // (timing experiments, tools/runs/r4_lc22_exp.sh: bit 2 leaves out the coefficient rows, bit 1 the library rows)
                if (DEJAVU_EXP22 & 4) continue;
#endif
                lds_dma_16(li < n_mine ? coef4 + h * pass16 + ((long long)k * 4 + lw) * 64 + lane : hot, dst);
            }
#pragma unroll
            for (int i = 0; i < SK * TL; ++i) {                         // library row (K-step kb + kk, slot lw + 4 t)
#ifdef DEJAVU_EXP22
                if (DEJAVU_EXP22 & 2) continue;
#endif
                const int kk = i / TL, t = i % TL;
                int k = kb + kk;
                k = k < NKT ? k : NKT - 1;
                const unsigned dst = __builtin_amdgcn_readfirstlane(slot + (unsigned)((COEF_ROWS + kk * KLIB + lw + 4 * t) * 1024));
                if (!llive[t]) lds_dma_16(hot, dst);
                else if (nt_rows) lds_dma_16_nt(reinterpret_cast<const uint4*>(lp[t] + (long long)k * 1024), dst);
                else lds_dma_16(reinterpret_cast<const uint4*>(lp[t] + (long long)k * 1024), dst);
            }
            lslot = lslot + 1 == RD ? 0 : lslot + 1;
            if (++lst == nst) { lst = 0; ++li; loader_item(); }
        };
        loader_item();
#pragma unroll
        for (int r = 0; r < RD - 1; ++r) issue_stage();
        for (long long j = 0; j < n_mine; ++j) {
            for (int st = 0; st < nst; ++st) {
#ifdef DEJAVU_EXP22
                wait_vmcnt_le<(((DEJAVU_EXP22 & 4) ? 0 : SK * HT) + ((DEJAVU_EXP22 & 2) ? 0 : SK * TL)) * (RD - 2)>();
#else
                wait_vmcnt_le<PER * (RD - 2)>();                        // this wave's rows of stage (j, st) have landed ...
#endif
                __builtin_amdgcn_s_barrier();                           // ... everybody's have; nobody still reads the slot before it
                issue_stage();                                          // (may belong to the next item: its pipeline fill)
            }
#pragma unroll
            for (int h = 0; h < HT; ++h)                                // the barriers of the consumers' finishing (three per call)
                if (a_off + 32 * h < fz.A_real) fused_finish_idle();
        }
        wait_vmcnt_le<0>();                                             // the re-reads past the last item
    } else {
        // ---- consumers
        int cslot = 0;                                                  // ring slot of the current stage
        int nfin = 0;                                                   // fused_finish calls so far (their parity)
        int hconst[HT][2];
#pragma unroll
        for (int h = 0; h < HT; ++h) {
            const int a = a_off + 32 * h + (lane & 31), ac = a < fz.A_real ? a : fz.A_real - 1;
            hconst[h][0] = acc_sum(fz.hsconst, ac);
            hconst[h][1] = fz.vconst ? acc_sum(fz.vconst, ac) : 0;
        }
        // widths of the two accumulator classes, per segment: position 0 | positions 1, 2, 3 (equal where they stand for something)
        auto w_of = [&](int seg, int cls) -> int {
            if (cls == 0) return b.wacc[seg][0];
            const int w1 = b.wacc[seg][1], w2 = b.wacc[seg][2], w3 = b.wacc[seg][3];
            return w1 ? w1 : (w2 ? w2 : w3);
        };
        const int whs = w_of(0, 0) ? w_of(0, 0) : w_of(0, 1);          // the saturation segment's one width
        for (long long j = 0; j < n_mine; ++j) {
            const long long item = blockIdx.x + j * gridDim.x;
            const long long g0 = (item * G32) / GQ, g1 = ((item + 1) * G32) / GQ;
            long long gidx[TL];
            bool live[TL];
#pragma unroll
            for (int t = 0; t < TL; ++t) {
                const long long g = g0 + wave * TL + t;
                live[t] = g < g1;
                gidx[t] = live[t] ? g : g0;
            }
            int tot_v[TL][HT][16];
            {
                v16f_t acc[TL][HT][2];
                auto clear = [&]() {
#pragma unroll
                    for (int t = 0; t < TL; ++t)
#pragma unroll
                        for (int h = 0; h < HT; ++h)
#pragma unroll
                            for (int cl = 0; cl < 2; ++cl)
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[t][h][cl][r] = 0.f;
                };
                clear();
                // operand registers: unit q = (K-step q / 2, heading tile q % 2) reads its four coefficient rows into a[q & 1]; the
                // two library rows of a K-step arrive with its first unit in xl[k & 1], and their masked operands bo are made once
                v4u_t a[2][4], xl[2][TL];
                auto fetch = [&](int slot_i, auto qc) {
                    constexpr int q = decltype(qc)::value;
                    constexpr int k = q / HT, h = q % HT;
                    const unsigned sad = lds_base + (unsigned)slot_i * (unsigned)SLOTB + (unsigned)lane * 16u;
#ifdef DEJAVU_EXP22
                    if ((DEJAVU_EXP22 & 16) && q > 0) return;                     // bit 4: only the stage's first unit reads its operands
#endif
                    static_for<4>([&](auto sc) { constexpr int s_ = decltype(sc)::value; lds_read16<(k * KCOEF + h * 4 + s_) * 1024>(a[q & 1][s_], sad); });
                    if constexpr (h == 0) {
                        const unsigned lad = sad + (unsigned)(COEF_ROWS * 1024 + wave * TL * 1024);
                        static_for<TL>([&](auto tc) { constexpr int t_ = decltype(tc)::value; lds_read16<(k * KLIB + t_) * 1024>(xl[k & 1][t_], lad); });
                    }
                };
                __builtin_amdgcn_s_barrier();                           // stage (j, 0) is in LDS
                if (j == 0) DV_STAMP(1);
                fetch(cslot, IntC<0>{});
                auto run_stages = [&](int s0, int s1) {
                    for (int st = s0; st < s1; ++st) {
                        const int kb = st * SK;
                        const int nslot = cslot + 1 == RD ? 0 : cslot + 1;
                        unsigned bo[TL][4][4];                          // the K-step's masked library operands, made once for both heading tiles
                        static_for<NU>([&](auto qc) {
                            constexpr int q = decltype(qc)::value;
                            constexpr int k = q / HT, h = q % HT;
                            if constexpr (q + 1 < NU) {
                                fetch(cslot, IntC<q + 1>{});            // one unit ahead
                                constexpr int h1 = (q + 1) % HT;
                                lds_wait<4 + (h1 == 0 ? TL : 0)>();     // all but the newest unit's reads have landed
                            } else {
                                lds_wait<0>();                          // everything this wave will use of the slot is in registers
                                if (st + 1 < nst) {
                                    __builtin_amdgcn_s_barrier();       // stage st + 1 is in LDS; the loaders may refill slot st - 1 ... and,
                                    fetch(nslot, IntC<0>{});            //   one barrier later, this one
                                }
                            }
#pragma unroll
                            for (int s = 0; s < 4; ++s) lds_tie(a[q & 1][s]);
                            if constexpr (h == 0) {
#pragma unroll
                                for (int t = 0; t < TL; ++t) lds_tie(xl[k & 1][t]);
                                // every position's bit to the 1.0 bit of its nibble (E2M1 0010): positions 0 / 2 / 3 by a shift
                                const unsigned m = kb + k < NKT ? 0x22222222u : 0u;
#pragma unroll
                                for (int t = 0; t < TL; ++t) {
                                    const unsigned x[4] = {xl[k & 1][t].x, xl[k & 1][t].y, xl[k & 1][t].z, xl[k & 1][t].w};
#pragma unroll
                                    for (int d = 0; d < 4; ++d) {
#ifdef DEJAVU_EXP22
                                        if (DEJAVU_EXP22 & 8) { bo[t][0][d] = bo[t][1][d] = bo[t][2][d] = bo[t][3][d] = x[d]; continue; }   // bit 3: no masks
#endif
                                        bo[t][0][d] = (x[d] << 1) & m;
                                        bo[t][1][d] = x[d] & m;
                                        bo[t][2][d] = (x[d] >> 1) & m;
                                        bo[t][3][d] = (x[d] >> 2) & m;
                                    }
                                }
                            }
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const v4u_t& av = a[q & 1][s];
                                const v8i_t ao = v8i_t{(int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0};
#pragma unroll
                                for (int t = 0; t < TL; ++t) {
                                    const v8i_t bv = v8i_t{(int)bo[t][s][0], (int)bo[t][s][1], (int)bo[t][s][2], (int)bo[t][s][3], 0, 0, 0, 0};
#ifdef DEJAVU_EXP22
                                    if (DEJAVU_EXP22 & 1) {                       // bit 0: no MFMA (the operands are consumed by one vector operation each)
                                        acc[t][h][s ? 1 : 0][0] += __int_as_float(bv[0] ^ ao[0] ^ bv[1] ^ ao[1] ^ bv[2] ^ ao[2] ^ bv[3] ^ ao[3]);
                                        continue;
                                    }
#endif
                                    acc[t][h][s ? 1 : 0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bv, ao, acc[t][h][s ? 1 : 0], 4, 4, 0, 0, 0, 0);   // views x headings
                                }
                            }
                        });
                        cslot = nslot;
                    }
                };
                // saturation stages (when the library has that segment), then the value stages; the accumulators change hands at the boundary
                const int nst0 = has_hs_sum ? (NK0 / SK < nst ? NK0 / SK : nst) : 0;     // (NK0 is a whole number of stages)
                const int hs_end = (has_hs_sum && !(c.hasv && b.NK[1] > 0)) ? nst : nst0;
                run_stages(0, hs_end);
                if (hs_end > 0) {
                    if (hs_end < nst) {                                 // the next stage's first operands are on their way: let them land
                        lds_wait<0>();                                  // before any compiler-made LDS traffic (once per item)
#pragma unroll
                        for (int s = 0; s < 4; ++s) lds_tie(a[0][s]);
#pragma unroll
                        for (int t = 0; t < TL; ++t) lds_tie(xl[0][t]);
                    }
                    // the segment's counts (one width) into this consumer's LDS block: [t][h][r][lane]
#pragma unroll
                    for (int t = 0; t < TL; ++t)
#pragma unroll
                        for (int h = 0; h < HT; ++h)
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                park[((t * HT + h) * 16 + r) * 64 + lane] = (short)((int)acc[t][h][0][r] + (int)acc[t][h][1][r]);
                    if (hs_end < nst) clear();
                }
                if (hs_end < nst) {
                    run_stages(hs_end, nst);
                    const int w0 = w_of(1, 0), w1 = w_of(1, 1);
#pragma unroll
                    for (int t = 0; t < TL; ++t)
#pragma unroll
                        for (int h = 0; h < HT; ++h)
#pragma unroll
                            for (int r = 0; r < 16; ++r) tot_v[t][h][r] = __mul24(w0, (int)acc[t][h][0][r]) + __mul24(w1, (int)acc[t][h][1][r]);
                } else {
#pragma unroll
                    for (int t = 0; t < TL; ++t)
#pragma unroll
                        for (int h = 0; h < HT; ++h)
#pragma unroll
                            for (int r = 0; r < 16; ++r) tot_v[t][h][r] = 0;
                }
            }
            if (j == 0) DV_STAMP(2);                                   // (diagnostic builds: the first item's loop, sums and parking are done)
#pragma unroll
            for (int h = 0; h < HT; ++h) {
                if (j == 0) DV_STAMP(3 + h);                            // 3, 4: in front of the first / second heading tile's finishing
                auto of_hs = [&](int t, int r) -> int { return __mul24(whs, (int)park[((t * HT + h) * 16 + r) * 64 + lane]); };
                auto of_v = [&](int t, int r) -> int { return tot_v[t][h][r]; };
#ifdef DEJAVU_EXP22             // bit 5: the finishing's barriers only (the timing builds' sums are not worth finishing)
                if (DEJAVU_EXP22 & 32) {
                    if (a_off + 32 * h < fz.A_real) {
                        if (of_hs(0, 0) + of_v(1, 15) == 0x7fffffff) park[lane] = 1;       // (the sums stay live)
                        fused_finish_idle();
                    }
                    continue;
                }
#endif
                if (a_off + 32 * h < fz.A_real)                         // (uniform: a heading tile without headings has nothing to finish)
                    fused_finish<TL, NW, true>(of_hs, of_v, gidx, live, scratch0, c, fz, a_off + 32 * h, has_hs_sum, j, lane, wave, nfin++ & 1, NC,
                                               hconst[h], scratch0 + kFuseBlk + h * 64);
            }
            if (j == 0) DV_STAMP(5);                                    // behind both
        }
    }
#pragma unroll
    for (int h = 0; h < HT; ++h)
        if (a_off + 32 * h < fz.A_real) fused_block_end(scratch0 + h * 64, fz, c, a_off + 32 * h);
    (void)apad_total;
}

// The pass of 64 headings with that body; off-level patches take the int8 ring body per heading tile, as in k_sad_mfma_dual (the
// same ranges of eight view groups).  Fused finishing only: the passes of an ensemble step.
template <int SKL, int RDL>
__global__ void __launch_bounds__(512, 2)
k_sad_lc22(const uint4* __restrict__ btiles, const uint4* __restrict__ coef, const uint4* __restrict__ coef4, const unsigned* __restrict__ offlevel,
           LibCfg c, BitCfg b, int apad_total, int a_off, int has_hs_sum, FuseArgs fz, int n_gq) {
    const bool fp4 = __builtin_amdgcn_readfirstlane(*offlevel) == 0u;
    if (fp4) {
        sad_lc22_fp4<SKL, RDL>(btiles, coef4, c, b, apad_total, a_off, has_hs_sum, fz, n_gq);
        return;
    }
    const int NKT = b.NK[0] + b.NK[1];
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
        if (h > 0 && a_off + 32 * h >= apad_total) break;
        sad_ring_i8<4, 1, 2, true>(btiles, coef + (long long)h * NKT * 512, nullptr, c, b, 1, apad_total, a_off + 32 * h, has_hs_sum, fz, n_gq);
    }
}

// One launch, both forms: `offlevel` (k_patch_prep) says whether this step's patches allow the fp4 coefficients.  The loader /
// consumer body of the fp4 form reads ftiles (the code tiles when the library has them, else the bit tiles), everything else the
// bit tiles.
template <int SK8, int RD8, int SK4, int RD4, int TILES, bool FUSE, int SKL, int RDL, bool LCODE, int HT>
__global__ void __launch_bounds__(512, 2)
k_sad_mfma_dual(const uint4* __restrict__ btiles, const uint4* __restrict__ ftiles, const uint4* __restrict__ coef, const uint4* __restrict__ coef4,
                const unsigned* __restrict__ offlevel, int* __restrict__ part, LibCfg c, BitCfg b, int nchunk, int apad_total, int a_off,
                int has_hs_sum, FuseArgs fz, int n_gq) {
    // HT = 2: ONE launch covers the 64 headings at a_off (two heading tiles, coefficient images one pass apart); the bodies that
    // multiply 32 headings per pass run twice
    const bool fp4 = __builtin_amdgcn_readfirstlane(*offlevel) == 0u;
    if constexpr (SKL > 0) {
        static_assert(TILES == 1, "the loader / consumer body cuts the library into ranges of 8 / HT view groups; the other bodies must agree");
        if (fp4 && nchunk == 1) {
            sad_lc_fp4<SKL, RDL, FUSE, LCODE, HT>(ftiles, coef4, part, c, b, apad_total, a_off, has_hs_sum, fz, n_gq);     // LCODE == (b.vcode != 0)
            return;
        }
    }
    const int NKT = b.NK[0] + b.NK[1];
#pragma unroll 1
    for (int h = 0; h < HT; ++h) {
        if (h > 0 && a_off + 32 * h >= apad_total) break;
        if (fp4) sad_ring_fp4<SK4, TILES, RD4, FUSE>(btiles, coef4 + (long long)h * NKT * 256, part, c, b, nchunk, apad_total, a_off + 32 * h, has_hs_sum, fz, n_gq);
        else sad_ring_i8<SK8, TILES, RD8, FUSE>(btiles, coef + (long long)h * NKT * 512, part, c, b, nchunk, apad_total, a_off + 32 * h, has_hs_sum, fz, n_gq);
    }
}


// Streaming-read microbenchmark: the scoring kernels' access pattern without their arithmetic -- every wave pulls 1-KB
// rows straight into LDS by non-temporal LDS-DMA, sixteen rows in flight per wave, eight waves per workgroup, one workgroup
// per CU (tools/exp/dma_rate.hip: 5.7 TB/s with eight in flight, 6.7 with sixteen) -- as the measured ceiling the library
// stream is held against.
constexpr int kStreamRows = 16;
__global__ void __launch_bounds__(512)
k_stream_read(const uint4* __restrict__ src, long long n16, unsigned* __restrict__ sink) {
    extern __shared__ uint4 lds_stream[];          // [8 waves][kStreamRows rows][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned base = (unsigned)(unsigned long long)(lds_ptr_t)lds_stream + (unsigned)wave * (unsigned)kStreamRows * 1024u;
    const long long rows = n16 / 64;
    const long long stride = (long long)gridDim.x * 8;
    long long row = (long long)blockIdx.x * 8 + wave;
    for (; row + (kStreamRows - 1) * stride < rows; row += kStreamRows * stride) {
#pragma unroll
        for (int i = 0; i < kStreamRows; ++i) lds_dma_16_nt(src + (row + i * stride) * 64 + lane, __builtin_amdgcn_readfirstlane(base + (unsigned)i * 1024u));
        wait_vmcnt_le<0>();
    }
    for (; row < rows; row += stride) {
        lds_dma_16_nt(src + row * 64 + lane, __builtin_amdgcn_readfirstlane(base));
        wait_vmcnt_le<0>();
    }
    if (lds_stream[threadIdx.x].x == 0x9E3779B9u && lds_stream[threadIdx.x].y == 0x12345u) sink[0] = 1;
}

}  // namespace dv
