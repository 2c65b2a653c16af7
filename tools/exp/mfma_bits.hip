// mfma_bits.hip -- probe for the bit-plane int8 MFMA form of the SAD scoring (timing + exactness; not product code).
//
// |a - b| for a library byte b drawn from a small level set is linear in b's thermometer bits B_t:
//     |a - b| = const(a) + sum_t B_t * (w_t - 2*clamp(a - l_t, 0, w_t))
// so sum over pixels and planes of (int8 coefficient) x (library bit) is an exact int8 GEMM: M = headings,
// N = views, K = pixels x planes.  The library is stored as BITS (one dwordx4 per lane and K-step = 256 K-elements of
// a 32-view group); slice s of a K-step is the MFMA whose B operand is (x_j & (0x01010101 << s)), j = 0..3: bytes of
// value 2^s * bit, one v_and_b32 per operand dword, and the 2^s is divided out of accumulator s at the very end.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_bits mfma_bits.hip && ./mfma_bits [views] [ksteps] [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__host__ __device__ inline unsigned long long mix(unsigned long long z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// library dword (view f, K-step ks, half h, dword j)
__host__ __device__ inline unsigned lib_word(long long f, int ks, int h, int j) {
    return (unsigned)mix(((unsigned long long)f * 4096ull + (unsigned long long)ks) * 8ull + (unsigned long long)(h * 4 + j) + 0x1234567ull);
}
// coefficient (heading m, K-step ks, half h, dword j, bit beta) in [-64, 64]
__host__ __device__ inline int coef_val(int m, int ks, int h, int j, int beta) {
    const unsigned long long z = mix((((unsigned long long)m * 4096ull + (unsigned long long)ks) * 8ull + (unsigned long long)(h * 4 + j)) * 32ull + (unsigned long long)beta + 0x9999ull);
    return (int)(z % 129ull) - 64;
}

__global__ void k_fill_lib(uint4* bt, long long G, int NK, int GS) {
    const long long total = G * NK * 64;
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(t & 63);
        const long long r = t >> 6;
        const int ks = (int)(r % NK);
        const long long g = r / NK;
        const long long f = g * 32 + (lane & 31);
        const int h = lane >> 5;
        bt[(g * GS + ks) * 64 + lane] = make_uint4(lib_word(f, ks, h, 0), lib_word(f, ks, h, 1), lib_word(f, ks, h, 2), lib_word(f, ks, h, 3));
    }
}
// coef[ks][s][lane] : uint4, dword j byte b = C(m = lane&31, ks, h = lane>>5, j, beta = s + 8b)
__global__ void k_fill_coef(uint4* cf, int NK) {
    const int total = NK * 8 * 64;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
        const int lane = t & 63, s = (t >> 6) & 7, ks = t >> 9;
        unsigned w[4];
        for (int j = 0; j < 4; ++j) {
            w[j] = 0;
            for (int b = 0; b < 4; ++b) w[j] |= ((unsigned)(coef_val(lane & 31, ks, lane >> 5, j, s + 8 * b) & 0xff)) << (8 * b);
        }
        cf[t] = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

__device__ __forceinline__ v4u load_nt(const uint4* p) { return __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p)); }

// One workgroup = NW waves; wave w scores view group (32 views) NW*item + w over K-steps [k0, k1) of chunk ch.
//   * coefficients (A operands) of a stage (SK K-steps, SK*8 KB) are staged global -> registers -> LDS one stage ahead,
//     two LDS buffers, one barrier per stage;
//   * library bits (B operands) are loaded two stages ahead into a register ring (HBM latency);
//   * A operands are read from LDS one slice-round ahead (a[s] is re-loaded right after the MFMA that used it);
//   * 4 accumulators: slices s and s+4 both carry the factor 2^(s&3) once x is shifted right by 4 for the upper four.
// MODE 0: full.  MODE 2: no ANDs.  MODE 3: no LDS re-reads of A.  MODE 4: no MFMA (everything else).
template <int NW, int SK, int MODE, int WPS>
__global__ void __launch_bounds__(64 * NW, WPS)
k_bits(const uint4* __restrict__ bt, const uint4* __restrict__ cf, int* __restrict__ out, long long G, int NK, int nchunk,
       long long Fpad, int GS) {
    extern __shared__ uint4 lds[];                // [2][SK][8][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long GQ = (G + NW - 1) / NW;
    const long long n_items = GQ * nchunk;
    constexpr int STAGE16 = SK * 8 * 64;          // uint4 per stage
    constexpr int PER_T = STAGE16 / (64 * NW);    // uint4 per thread per stage
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ch = (int)(item / GQ);
        const long long gq = item - (long long)ch * GQ;
        long long g = gq * NW + wave;
        const bool live = g < G;
        if (!live) g = G - 1;
        const int k0 = (int)(((long long)ch * NK) / nchunk), k1 = (int)(((long long)(ch + 1) * NK) / nchunk);
        const int nst = (k1 - k0 + SK - 1) / SK;
        const uint4* lib = bt + (g * GS) * 64 + lane;
        v16i acc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][r] = 0;
        v4u ring[3][SK];
        uint4 creg[PER_T];
        auto kclamp = [&](int k) { return k < k1 ? k : k1 - 1; };
        // prologue: coefficients of stage 0 into registers, library bits of stages 0 and 1 into the ring
#pragma unroll
        for (int i = 0; i < PER_T; ++i) creg[i] = cf[(long long)k0 * 512 + i * (64 * NW) + threadIdx.x];
#pragma unroll
        for (int k = 0; k < SK; ++k) ring[0][k] = load_nt(lib + (long long)kclamp(k0 + k) * 64);
#pragma unroll
        for (int k = 0; k < SK; ++k) ring[1][k] = load_nt(lib + (long long)kclamp(k0 + SK + k) * 64);
        for (int st0 = 0; st0 < nst; st0 += 3) {
#pragma unroll
            for (int u = 0; u < 3; ++u) {           // ring slot of stage st is st % 3: unrolled so that it is a register name
                const int st = st0 + u;
                if (st < nst) {
                    const int kb = k0 + st * SK;
                    // this stage's coefficients (loaded one stage ago; the value is carried around the loop, so the
                    // compiler cannot sink the loads down to this use) -> LDS; buffer st&1 was last read in stage st-2
                    if (!(MODE & 32)) {
                        uint4* nb = lds + (st & 1) * STAGE16;
#pragma unroll
                        for (int i = 0; i < PER_T; ++i) nb[i * (64 * NW) + threadIdx.x] = creg[i];
                        __syncthreads();
                    }
                    {
#pragma unroll
                        for (int i = 0; i < PER_T; ++i) {
                            long long idx = (long long)(kb + SK) * 512 + i * (64 * NW) + threadIdx.x;
                            const long long lim = (long long)k1 * 512;
                            if (idx >= lim) idx = lim - 1;
                            if (!(MODE & 16)) creg[i] = cf[idx];
                        }
                    }
                    if (!(MODE & 1)) {
#pragma unroll
                        for (int k = 0; k < SK; ++k) ring[(u + 2) % 3][k] = load_nt(lib + (long long)kclamp(kb + 2 * SK + k) * 64);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const uint4* buf = lds + (st & 1) * STAGE16 + lane;
                    v4i a[2][8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) { const uint4 t = buf[s * 64]; a[0][s] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w}; }
#pragma unroll
                    for (int k = 0; k < SK; ++k) {
                        // all eight A operands of the next K-step are requested before this K-step's MFMAs start
                        if (!(MODE & 4) && k + 1 < SK) {
#pragma unroll
                            for (int s = 0; s < 8; ++s) {
                                const uint4 t = buf[((k + 1) * 8 + s) * 64];
                                a[(k + 1) & 1][s] = v4i{(int)t.x, (int)t.y, (int)t.z, (int)t.w};
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const v4u x = ring[u][k];
                        const v4u xs = v4u{x.x >> 4, x.y >> 4, x.z >> 4, x.w >> 4};
                        const bool on = kb + k < k1;
#pragma unroll
                        for (int s = 0; s < 8; ++s) {
                            const unsigned m = on ? (0x01010101u << (s & 3)) : 0u;
                            const v4u src = s < 4 ? x : xs;
                            v4i b;
                            if (MODE & 2) b = v4i{(int)src.x, (int)src.y, (int)src.z, (int)src.w};
                            else b = v4i{(int)(src.x & m), (int)(src.y & m), (int)(src.z & m), (int)(src.w & m)};
                            const v4i av = (MODE & 4) ? a[0][s] : a[k & 1][s];
                            if (!(MODE & 8)) acc[s & 3] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, b, acc[s & 3], 0, 0, 0);
                            else acc[s & 3][0] += av.x ^ b.x ^ av.y ^ b.y ^ av.z ^ b.z ^ av.w ^ b.w;
                        }
                    }
                }
            }
        }
        // fold the slices: accumulator s holds 2^s x its sum (s = 0..3)
        if (live) {
            int* dst = out + ((long long)ch * 32) * Fpad + g * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tot = acc[0][r] + (acc[1][r] >> 1) + (acc[2][r] >> 2) + (acc[3][r] >> 3);
                const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                dst[(long long)m * Fpad] = tot;
            }
        }
        __syncthreads();
    }
}

// Second structure: coefficients go global -> LDS by LDS-DMA (no staging registers, no ds_write), one barrier per stage,
// library bits one stage ahead (cur + next register sets), TILES view groups per wave sharing every A operand.
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;
template <int NW, int SK, int TILES, int WPS>
__global__ void __launch_bounds__(64 * NW, WPS)
k_bits2(const uint4* __restrict__ bt, const uint4* __restrict__ cf, int* __restrict__ out, long long G, int NK, int nchunk,
        long long Fpad, int GS) {
    extern __shared__ uint4 lds[];                // [2][SK][8][64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int VW = NW * TILES;                // view groups per workgroup item
    const long long GQ = (G + VW - 1) / VW;
    const long long n_items = GQ * nchunk;
    constexpr int STAGE16 = SK * 8 * 64;          // uint4 per stage
    constexpr int PER_W = STAGE16 / (64 * NW);    // 1 KB rows per wave per stage
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ch = (int)(item / GQ);
        const long long gq = item - (long long)ch * GQ;
        const int k0 = (int)(((long long)ch * NK) / nchunk), k1 = (int)(((long long)(ch + 1) * NK) / nchunk);
        const int nst = (k1 - k0 + SK - 1) / SK;
        const uint4* lib[TILES];
        bool live[TILES];
        long long gidx[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            long long g = gq * VW + wave * TILES + t;
            live[t] = g < G;
            if (!live[t]) g = G - 1;
            gidx[t] = g;
            lib[t] = bt + (g * GS) * 64 + lane;
        }
        v16i acc[TILES][4];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][s][r] = 0;
        v4u ring[2][TILES][SK];
        auto kclamp = [&](int k) { return k < k1 ? k : k1 - 1; };
        auto dma_stage = [&](int st) {            // coefficient rows of stage st -> LDS buffer st & 1
            const int kb = k0 + st * SK;
#pragma unroll
            for (int i = 0; i < PER_W; ++i) {
                const int row = wave * PER_W + i;                     // 1 KB row of the stage: (K-step, slice)
                long long src = (long long)kb * 512 + row * 64 + lane;
                const long long lim = (long long)k1 * 512;
                if (src >= lim) src = lim - 64 + lane;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(cf + src), (lds_ptr_t)(lds + (st & 1) * STAGE16 + row * 64), 16, 0, 0);
            }
        };
        dma_stage(0);
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int k = 0; k < SK; ++k) ring[0][t][k] = load_nt(lib[t] + (long long)kclamp(k0 + k) * 64);
        for (int st0 = 0; st0 < nst; st0 += 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int st = st0 + u;
                if (st < nst) {
                    const int kb = k0 + st * SK;
                    __builtin_amdgcn_s_waitcnt(0x0f70);          // vmcnt(0): this wave's DMA rows and library bits have landed
                    __syncthreads();                              // everybody's have; and everybody is done reading the other buffer
                    if (st + 1 < nst) dma_stage(st + 1);
#pragma unroll
                    for (int t = 0; t < TILES; ++t)
#pragma unroll
                        for (int k = 0; k < SK; ++k) ring[u ^ 1][t][k] = load_nt(lib[t] + (long long)kclamp(kb + SK + k) * 64);
                    __builtin_amdgcn_sched_barrier(0);
                    const uint4* buf = lds + (st & 1) * STAGE16 + lane;
                    // half-steps: slices 0..3 of a K-step use x, slices 4..7 use x >> 4; the four A operands of the next
                    // half-step are requested before this half-step's MFMAs start
                    v4i a[2][4];
#pragma unroll
                    for (int s = 0; s < 4; ++s) { const uint4 w = buf[s * 64]; a[0][s] = v4i{(int)w.x, (int)w.y, (int)w.z, (int)w.w}; }
#pragma unroll
                    for (int hs = 0; hs < 2 * SK; ++hs) {
                        if (hs + 1 < 2 * SK) {
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const uint4 w = buf[((hs + 1) * 4 + s) * 64];
                                a[(hs + 1) & 1][s] = v4i{(int)w.x, (int)w.y, (int)w.z, (int)w.w};
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        const int k = hs >> 1;
                        const bool on = kb + k < k1;
#pragma unroll
                        for (int t = 0; t < TILES; ++t) {
                            const v4u x = ring[u][t][k];
                            const v4u src = (hs & 1) ? v4u{x.x >> 4, x.y >> 4, x.z >> 4, x.w >> 4} : x;
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const unsigned m = on ? (0x01010101u << s) : 0u;
                                const v4i b = v4i{(int)(src.x & m), (int)(src.y & m), (int)(src.z & m), (int)(src.w & m)};
                                acc[t][s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[hs & 1][s], b, acc[t][s], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            if (live[t]) {
                int* dst = out + ((long long)ch * 32) * Fpad + gidx[t] * 32 + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tot = acc[t][0][r] + (acc[t][1][r] >> 1) + (acc[t][2][r] >> 2) + (acc[t][3][r] >> 3);
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    dst[(long long)m * Fpad] = tot;
                }
            }
        }
        __syncthreads();
    }
}

static int g_gs = 0;
template <int NW, int SK, int MODE, int WPS>
static float run(const uint4* bt, const uint4* cf, int* out, long long G, int NK, int nchunk, long long Fpad, int wg_per_cu, int reps) {
    const int GS = g_gs;
    const size_t lds = (size_t)2 * SK * 8 * 64 * 16;
    CHECK(hipFuncSetAttribute((const void*)k_bits<NW, SK, MODE, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long items = ((G + NW - 1) / NW) * nchunk;
    long long grid = 256ll * wg_per_cu;
    if (grid > items) grid = items;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_bits<NW, SK, MODE, WPS>), dim3((unsigned)grid), dim3(64 * NW), lds, 0, bt, cf, out, G, NK, nchunk, Fpad, GS);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((k_bits<NW, SK, MODE, WPS>), dim3((unsigned)grid), dim3(64 * NW), lds, 0, bt, cf, out, G, NK, nchunk, Fpad, GS);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

template <int NW, int SK, int TILES, int WPS>
static float run2(const uint4* bt, const uint4* cf, int* out, long long G, int NK, int nchunk, long long Fpad, int wg_per_cu, int reps) {
    const int GS = g_gs;
    const size_t lds = (size_t)2 * SK * 8 * 64 * 16;
    CHECK(hipFuncSetAttribute((const void*)k_bits2<NW, SK, TILES, WPS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long items = ((G + NW * TILES - 1) / (NW * TILES)) * nchunk;
    long long grid = 256ll * wg_per_cu;
    if (grid > items) grid = items;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_bits2<NW, SK, TILES, WPS>), dim3((unsigned)grid), dim3(64 * NW), lds, 0, bt, cf, out, G, NK, nchunk, Fpad, GS);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((k_bits2<NW, SK, TILES, WPS>), dim3((unsigned)grid), dim3(64 * NW), lds, 0, bt, cf, out, G, NK, nchunk, Fpad, GS);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

static int check(const int* out, int nchunk, long long F, long long Fpad, int NK, const char* what) {
    std::vector<int> h((size_t)nchunk * 32 * Fpad);
    CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0, checked = 0;
    const long long probe_f[8] = {0, 1, 31, 32, 63, 64, F / 2 + 7, F - 1};
    for (long long f : probe_f)
        for (int m = 0; m < 32; m += 5) {
            long long want = 0;
            for (int ks = 0; ks < NK; ++ks)
                for (int hh = 0; hh < 2; ++hh)
                    for (int j = 0; j < 4; ++j) {
                        const unsigned w = lib_word(f, ks, hh, j);
                        for (int b = 0; b < 32; ++b) if ((w >> b) & 1) want += coef_val(m, ks, hh, j, b);
                    }
            long long got = 0;
            for (int ch = 0; ch < nchunk; ++ch) got += h[((size_t)ch * 32 + m) * Fpad + f];
            ++checked;
            if (got != want) { if (bad < 4) printf("MISMATCH f=%lld m=%d got %lld want %lld\n", f, m, got, want); ++bad; }
        }
    printf("exactness (%s): %d of %d sampled sums wrong\n", what, bad, checked);
    return bad;
}

int main(int argc, char** argv) {
    const long long F = argc > 1 ? atoll(argv[1]) : 500000;
    const int NK = argc > 2 ? atoi(argv[2]) : 384;          // 128x128 px x 6 planes / 256
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    const int GS = argc > 4 ? atoi(argv[4]) : (NK | 1);
    g_gs = GS;
    const long long G = (F + 31) / 32, Fpad = G * 32;
    uint4 *bt, *cf;
    int* out;
    const int max_chunk = 8;
    CHECK(hipMalloc(&bt, (size_t)G * GS * 1024));
    CHECK(hipMalloc(&cf, (size_t)NK * 8 * 1024));
    CHECK(hipMalloc(&out, (size_t)max_chunk * 32 * Fpad * 4));
    hipLaunchKernelGGL(k_fill_lib, dim3(4096), dim3(256), 0, 0, bt, G, NK, GS);
    hipLaunchKernelGGL(k_fill_coef, dim3(256), dim3(256), 0, 0, cf, NK);
    CHECK(hipDeviceSynchronize());
    const double lib_gb = (double)G * NK * 1024 / 1e9;
    const double mfmas = (double)G * NK * 8;
    printf("group stride %d KB; ", GS); printf("views %lld  K-steps %d  library %.3f GB  MFMAs %.3g (%.1f us at 32 cyc x 1024 SIMDs x 2.4 GHz)\n", F, NK, lib_gb, mfmas,
           mfmas * 32 / 1024 / 2.4e3);

    // exactness: nchunk = 2, full kernel
    {
        const int nchunk = 2;
        run<4, 4, 0, 2>(bt, cf, out, G, NK, nchunk, Fpad, 2, 1);
        std::vector<int> h((size_t)nchunk * 32 * Fpad);
        CHECK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
        int bad = 0, checked = 0;
        const long long probe_f[6] = {0, 1, 31, 32, F / 2 + 7, F - 1};
        for (long long f : probe_f)
            for (int m = 0; m < 32; m += 5) {
                long long want = 0;
                for (int ks = 0; ks < NK; ++ks)
                    for (int hh = 0; hh < 2; ++hh)
                        for (int j = 0; j < 4; ++j) {
                            const unsigned w = lib_word(f, ks, hh, j);
                            for (int b = 0; b < 32; ++b) if ((w >> b) & 1) want += coef_val(m, ks, hh, j, b);
                        }
                long long got = 0;
                for (int ch = 0; ch < nchunk; ++ch) got += h[((size_t)ch * 32 + m) * Fpad + f];
                ++checked;
                if (got != want) { if (bad < 8) printf("MISMATCH f=%lld m=%d got %lld want %lld\n", f, m, got, want); ++bad; }
            }
        printf("exactness: %d of %d sampled sums wrong\n", bad, checked);
    }
    struct { const char* name; float ms; } res[16];
    int n = 0;
    res[n++] = {"v1 NW8 SK4 full", run<8, 4, 0, 2>(bt, cf, out, G, NK, 1, Fpad, 1, reps)};
    CHECK(hipMemset(out, 0, (size_t)2 * 32 * Fpad * 4));
    run2<8, 4, 1, 2>(bt, cf, out, G, NK, 2, Fpad, 1, 1); check(out, 2, F, Fpad, NK, "dma NW8 SK4 T1 chunk2");
    CHECK(hipMemset(out, 0, (size_t)2 * 32 * Fpad * 4));
    run2<8, 4, 2, 2>(bt, cf, out, G, NK, 1, Fpad, 1, 1); check(out, 1, F, Fpad, NK, "dma NW8 SK4 T2");
    res[n++] = {"dma NW8 SK4 T1 1wg", run2<8, 4, 1, 2>(bt, cf, out, G, NK, 1, Fpad, 1, reps)};
    res[n++] = {"dma NW8 SK8 T1 1wg", run2<8, 8, 1, 2>(bt, cf, out, G, NK, 1, Fpad, 1, reps)};
    res[n++] = {"dma NW8 SK4 T2 1wg", run2<8, 4, 2, 2>(bt, cf, out, G, NK, 1, Fpad, 1, reps)};
    res[n++] = {"dma NW8 SK2 T2 1wg", run2<8, 2, 2, 2>(bt, cf, out, G, NK, 1, Fpad, 1, reps)};
    res[n++] = {"dma NW8 SK8 T2 1wg", run2<8, 8, 2, 2>(bt, cf, out, G, NK, 1, Fpad, 1, reps)};
    res[n++] = {"dma NW4 SK4 T2 2wg", run2<4, 4, 2, 2>(bt, cf, out, G, NK, 1, Fpad, 2, reps)};
    res[n++] = {"dma NW8 SK4 T1 1wg chunk3", run2<8, 4, 1, 2>(bt, cf, out, G, NK, 3, Fpad, 1, reps)};
    for (int i = 0; i < n; ++i)
        printf("%-52s %8.3f ms  %6.2f TB/s of library bits  %5.1f cyc/MFMA/SIMD at 2.4 GHz\n", res[i].name, res[i].ms,
               lib_gb / res[i].ms, res[i].ms * 1e-3 * 2.4e9 * 1024 / mfmas);
    return 0;
}
