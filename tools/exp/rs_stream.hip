// What would a scoring body cost that takes the library rows OUT of the LDS?  ("the consumers".)
//   hipcc --offload-arch=gfx950 -O3 -o rs_stream rs_stream.hip && ./rs_stream
// One wave per SIMD (4 waves per workgroup, one workgroup per CU, up to 512 registers per wave).  Every wave streams the 1-KB rows of its
// own two "view groups" straight from HBM into a ring of D K-steps of PINNED registers (inline asm: the compiler never sees a register
// that has a load in flight), one counted s_waitcnt per K-step, and multiplies each row four times on the matrix cores
// (v_mfma_f32_32x32x64_f8f6f4, the masks of sad_lc_fp4 in front of every MFMA) against coefficient operands read from LDS (static
// contents here: 4 ds_read_b128 per K-step, as the consumers of sad_lc_fp4 read them).  Results are meaningless; the time per K-step and
// the stream rate are the point.  Printed: ns per K-step per wave and TB/s over the chip, with and without the masks / the MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define RING_LO 120          // ring = v[120:247]: 16 K-steps x 2 rows x 4 registers
#define CLOB "v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139", \
 "v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159", \
 "v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179", \
 "v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191","v192","v193","v194","v195","v196","v197","v198","v199", \
 "v200","v201","v202","v203","v204","v205","v206","v207","v208","v209","v210","v211","v212","v213","v214","v215","v216","v217","v218","v219", \
 "v220","v221","v222","v223","v224","v225","v226","v227","v228","v229","v230","v231","v232","v233","v234","v235","v236","v237","v238","v239", \
 "v240","v241","v242","v243","v244","v245","v246","v247"

template <int U, int T>          // row (slot U, tile T) <- 1 KB at base + off
__device__ __forceinline__ void ring_load(unsigned voff, const void* base, int) {
    constexpr int R = RING_LO + (U * 2 + T) * 4;
    asm volatile("global_load_dwordx4 v[%0:%1], %2, %3 nt" :: "n"(R), "n"(R + 3), "v"(voff), "s"(base) : "memory", CLOB);
}
template <int U, int T, int DW>  // (ring dword >> SH) & mask -> a compiler register
__device__ __forceinline__ unsigned ring_and(unsigned mask, int sh) {
    constexpr int R = RING_LO + (U * 2 + T) * 4 + DW;
    unsigned o;
    if (sh) asm volatile("v_lshrrev_b32 %0, 2, v[%1]\n\tv_and_b32 %0, %2, %0" : "=v"(o) : "n"(R), "s"(mask) : CLOB);
    else asm volatile("v_and_b32 %0, %2, v[%1]" : "=v"(o) : "n"(R), "s"(mask) : CLOB);
    return o;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory", CLOB); }

template <int U> struct IC { static constexpr int v = U; };
template <int N, int I = 0, typename F> __device__ __forceinline__ void sfor(F&& f) { if constexpr (I < N) { f(IC<I>{}); sfor<N, I + 1>(f); } }

template <int D, bool MASKS, bool MFMA>
__global__ void __launch_bounds__(256, 1) k(const unsigned char* lib, long long rows_per_wave, unsigned long long* out, const unsigned* seedp) {
    static_assert(D == 16, "ring of 16 K-steps");
    __shared__ v4u coef[8 * 4 * 64];                               // 8 K-steps x 4 bit positions x 64 lanes
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 8 * 4 * 64; i += 256) coef[i] = v4u{seedp[i & 255], seedp[(i + 7) & 255], seedp[(i + 3) & 255], seedp[(i + 11) & 255]};
    __syncthreads();
    v16f acc[2][4];
    for (int t = 0; t < 2; ++t) for (int s = 0; s < 4; ++s) for (int r = 0; r < 16; ++r) acc[t][s][r] = 0.f;
    const long long gw = (long long)blockIdx.x * 4 + wave;        // this wave's two streams of rows_per_wave rows of 1 KB each
    const unsigned char* b0 = lib + (gw * 2 + 0) * rows_per_wave * 1024;
    const unsigned char* b1 = lib + (gw * 2 + 1) * rows_per_wave * 1024;
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned m0 = seedp[300] | 0x11111111u;
    // fill the ring
    sfor<D>([&](auto uc) { constexpr int u = decltype(uc)::v; ring_load<u, 0>(voff, b0 + u * 1024, 0); ring_load<u, 1>(voff, b1 + u * 1024, 0); });
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (long long k0 = 0; k0 + D <= rows_per_wave; k0 += D) {
        sfor<D>([&](auto uc) {
            constexpr int u = decltype(uc)::v;
            wait_vm<2 * (D - 1)>();                                // K-step k0 + u has landed: the 15 younger K-steps' 30 loads may be out
            const v4u* crow = &coef[((u & 7) * 4) * 64 + lane];
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    unsigned o[4];
                    const unsigned m = s < 3 ? (m0 << s) : (m0 << 1);
                    if (MASKS) {
                        if (t == 0) { o[0] = ring_and<u, 0, 0>(m, s == 3); o[1] = ring_and<u, 0, 1>(m, s == 3); o[2] = ring_and<u, 0, 2>(m, s == 3); o[3] = ring_and<u, 0, 3>(m, s == 3); }
                        else { o[0] = ring_and<u, 1, 0>(m, s == 3); o[1] = ring_and<u, 1, 1>(m, s == 3); o[2] = ring_and<u, 1, 2>(m, s == 3); o[3] = ring_and<u, 1, 3>(m, s == 3); }
                    } else {
                        o[0] = o[1] = o[2] = o[3] = m;
                    }
                    if (MFMA) {
                        const v4u cv = crow[s * 64];
                        const v8i bv = v8i{(int)o[0], (int)o[1], (int)o[2], (int)o[3], 0, 0, 0, 0};
                        const v8i av = v8i{(int)cv.x, (int)cv.y, (int)cv.z, (int)cv.w, 0, 0, 0, 0};
                        acc[t][s] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bv, av, acc[t][s], 4, 4, 0, 0, 0, 0);
                    } else {
                        acc[t][s][0] += __uint_as_float(o[0] ^ o[1] ^ o[2] ^ o[3]);
                    }
                }
            }
            // the slot is read: K-step k0 + u + D into it (past the end: the last rows again)
            const long long kn = k0 + u + D < rows_per_wave ? k0 + u + D : rows_per_wave - 1;
            ring_load<u, 0>(voff, b0 + kn * 1024, 0);
            ring_load<u, 1>(voff, b1 + kn * 1024, 0);
        });
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory", CLOB);
    float sum = 0.f;
    for (int t = 0; t < 2; ++t) for (int s = 0; s < 4; ++s) for (int r = 0; r < 16; ++r) sum += acc[t][s][r];
    if (sum == 1.2345e-30f) out[2] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int D, bool MASKS, bool MFMA>
static void run(const unsigned char* lib, long long rows, unsigned long long* d_out, const unsigned* d_seed, const char* what) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<D, MASKS, MFMA>), dim3(256), dim3(256), 0, 0, lib, rows, d_out, d_seed);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<D, MASKS, MFMA>), dim3(256), dim3(256), 0, 0, lib, rows, d_out, d_seed);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long cyc = 0;
    (void)hipMemcpy(&cyc, d_out, 8, hipMemcpyDeviceToHost);
    const double bytes = 256.0 * 4 * 2 * rows * 1024;
    printf("%-44s kernel %.3f ms  %.2f TB/s  %.1f ns per K-step and wave (%.0f shader cycles)\n", what, ms, bytes / (ms * 1e-3) / 1e12, ms * 1e6 / rows,
           (double)cyc / (rows / 16 * 16));
}

int main() {
    const long long rows = 3072;                                    // per stream: 256 CUs x 4 waves x 2 streams x 3 MB = 6.4 GB
    unsigned char* lib; unsigned long long* d_out; unsigned* d_seed;
    if (hipMalloc(&lib, (size_t)256 * 4 * 2 * rows * 1024) != hipSuccess) { printf("no memory\n"); return 1; }
    (void)hipMemset(lib, 0x5a, (size_t)256 * 4 * 2 * rows * 1024);
    (void)hipMalloc(&d_out, 64); (void)hipMalloc(&d_seed, 4096);
    std::vector<unsigned> h(1024, 0x13579bdfu);
    (void)hipMemcpy(d_seed, h.data(), 4096, hipMemcpyHostToDevice);
    run<16, true, true>(lib, rows, d_out, d_seed, "stream + masks + MFMA (the whole body)");
    run<16, false, true>(lib, rows, d_out, d_seed, "stream + MFMA, no masks");
    run<16, true, false>(lib, rows, d_out, d_seed, "stream + masks, no MFMA");
    run<16, false, false>(lib, rows, d_out, d_seed, "stream alone");
    return 0;
}
