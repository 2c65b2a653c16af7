// Ladder probe for the fp4 scoring loop: what does each ingredient of the loop cost?  (Not part of the product.)
//   L0: MFMAs only, 8 accumulators per wave (2 view groups x 4 bit positions), operands in registers
//   L1: + the 4 v_and (+ shift) per MFMA that make the B operand
//   L2: + operands read from LDS every K-step (4 coefficient rows + 2 library rows, ds_read_b128)
//   L3: + a workgroup barrier every 2 K-steps
//   L4: + LDS-DMA of 5 rows per wave per 2 K-steps from a large buffer (the HBM stream), counted vmcnt
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef float v16f_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ void dma16(const uint4* g, unsigned lds) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(g), "s"(lds) : "memory");
}
template <int L, int NW, int TL>
__global__ void __launch_bounds__(NW * 64, TL == 4 ? 1 : 2) k(const uint4* src, long long rows, float* out, int ksteps) {
    extern __shared__ uint4 lds[];           // 3 slots x (8 + 4 NW) KB
    constexpr int SLOT = 8 + 2 * TL * NW;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lbase = (unsigned)(unsigned long long)(lds_ptr_t)lds;
    v16f_t acc[TL][4];
    for (int t = 0; t < TL; ++t) for (int s = 0; s < 4; ++s) for (int r = 0; r < 16; ++r) acc[t][s][r] = 0.f;
    uint4 a[4], x[TL];
    for (int s = 0; s < 4; ++s) a[s] = src[lane + 64 * s];
    for (int t = 0; t < TL; ++t) x[t] = src[lane + 64 * (4 + t)];
    long long row = ((long long)blockIdx.x * NW + wave) * 4096 % (rows - 8192);
    constexpr int ND = 2 * TL + 8 / NW;                        // DMA rows per wave and stage: 4 library rows + its share of 8 coefficient rows
    if (L >= 4) {                                        // prologue: two stages in flight
        for (int st = 0; st < 2; ++st)
            for (int d = 0; d < ND; ++d) dma16(src + (row + st * ND + d) * 64 + lane, __builtin_amdgcn_readfirstlane(lbase + (unsigned)((st * SLOT + wave * ND + d) * 1024)));
    }
    for (int ks = 0; ks < ksteps; ks += 2) {
        const int st = ks >> 1;
        if (L >= 4) { if (ND == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); else if (ND == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
        if (L >= 3) __builtin_amdgcn_s_barrier();
        if (L >= 4) {
            for (int d = 0; d < ND; ++d) dma16(src + (row + (st + 2) * ND + d) * 64 + lane, __builtin_amdgcn_readfirstlane(lbase + (unsigned)((((st + 2) % 3) * SLOT + wave * ND + d) * 1024)));
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (L >= 2) {
                const uint4* slot = lds + ((st % 3) * SLOT) * 64 + lane;
#pragma unroll
                for (int s = 0; s < 4; ++s) a[s] = slot[(k * 4 + s) * 64];
#pragma unroll
                for (int t = 0; t < TL; ++t) x[t] = slot[(8 + wave * 2 * TL + k * TL + t) * 64];
            }
#pragma unroll
            for (int t = 0; t < TL; ++t)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    v8i_t bo, ao = {(int)a[s].x, (int)a[s].y, (int)a[s].z, (int)a[s].w, 0, 0, 0, 0};
                    if (L >= 1) {
                        const unsigned m = (s < 3 ? (0x11111111u << s) : 0x22222222u) | (unsigned)ks;
                        const int sh = s < 3 ? 0 : 2;
                        bo = v8i_t{(int)((x[t].x >> sh) & m), (int)((x[t].y >> sh) & m), (int)((x[t].z >> sh) & m), (int)((x[t].w >> sh) & m), 0, 0, 0, 0};
                    } else {
                        bo = v8i_t{(int)x[t].x, (int)x[t].y, (int)x[t].z, (int)x[t].w, 0, 0, 0, 0};
                    }
                    acc[t][s] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bo, ao, acc[t][s], 4, 4, 0, 0, 0, 0);
                }
        }
    }
    if (L >= 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
    for (int t = 0; t < TL; ++t) for (int s = 0; s < 4; ++s) for (int r = 0; r < 16; ++r) sum += acc[t][s][r];
    if (sum == 12345.678f) out[0] = sum;
}
template <int L, int NW, int TL>
int run(const uint4* buf, long long rows, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ksteps = 384 * 4;                          // four items of the 500 000-view problem per CU
    const size_t l = 3 * (8 + 2 * TL * NW) * 1024;
    CK(hipFuncSetAttribute((const void*)k<L, NW, TL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<L, NW, TL>), dim3(256 * 16 / (NW * TL)), dim3(NW * 64), l, 0, buf, rows, out, ksteps);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep == 2) printf("NW=%d TL=%d L%d: %.3f ms for %d K-steps per CU (%.0f ns per K-step)\n", NW, TL, L, ms, ksteps, ms * 1e6 / ksteps);
    }
    return 0;
}
int main() {
    const size_t bytes = 6ull << 30;
    uint4* buf; float* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 16));
    CK(hipMemset(buf, 0x22, bytes));
    const long long rows = bytes / 1024;
    if (run<1, 8, 2>(buf, rows, out)) return 1;
    if (run<2, 8, 2>(buf, rows, out)) return 1;
    if (run<3, 8, 2>(buf, rows, out)) return 1;
    if (run<4, 8, 2>(buf, rows, out)) return 1;
    if (run<1, 4, 4>(buf, rows, out)) return 1;
    if (run<2, 4, 4>(buf, rows, out)) return 1;
    if (run<3, 4, 4>(buf, rows, out)) return 1;
    if (run<4, 4, 4>(buf, rows, out)) return 1;
    return 0;
}
