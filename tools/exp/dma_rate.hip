// Probe: is the LDS-DMA stream bound by bytes or by wave-instructions?  Rows of 1024 / 768 / 512 bytes (64 / 48 / 32 active
// lanes of global_load_lds_dwordx4 nt), eight in flight per wave, eight waves per workgroup, two workgroups per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int LANES>
__device__ __forceinline__ void dma(const unsigned char* g, unsigned lds) {
    unsigned keep; unsigned long long ke;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b32 m0, %3\n\ts_lshr_b64 exec, -1, %4\n\tglobal_load_lds_dwordx4 %2, off nt\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(ke) : "v"(g), "s"(lds), "n"(64 - LANES) : "memory", "scc");
}
template <int LANES, int INFL>
__global__ void __launch_bounds__(512) k(const unsigned char* src, long long rows, unsigned* sink) {
    extern __shared__ uint4 lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned base = (unsigned)(unsigned long long)(lds_ptr_t)lds + (unsigned)wave * INFL * 1024u;
    const long long stride = (long long)gridDim.x * 8;
    constexpr int RB = LANES * 16;
    long long row = (long long)blockIdx.x * 8 + wave;
    for (; row + (INFL - 1) * stride < rows; row += INFL * stride) {
#pragma unroll
        for (int i = 0; i < INFL; ++i) dma<LANES>(src + (row + i * stride) * RB + lane * 16, __builtin_amdgcn_readfirstlane(base + (unsigned)i * 1024u));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (lds[threadIdx.x].x == 0x9E3779B9u) sink[0] = 1;
}
template <int LANES, int INFL>
int run(const unsigned char* buf, size_t bytes, unsigned* sink) {
    const long long rows = bytes / (LANES * 16);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t l = 8 * INFL * 1024;
    CK(hipFuncSetAttribute((const void*)k<LANES, INFL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<LANES, INFL>), dim3(256 * (INFL > 8 ? 1 : 2)), dim3(512), l, 0, buf, rows, sink);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep) printf("rows of %4d B, %2d in flight per wave: %.3f ms per pass, %.2f TB/s, %.2f G rows/s\n", LANES * 16, INFL, ms / 5, bytes / (ms / 5 * 1e-3) / 1e12, rows / (ms / 5 * 1e-3) / 1e9);
    }
    return 0;
}
int main() {
    const size_t bytes = 3ull << 30;
    unsigned char* buf; unsigned* sink;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 16));
    CK(hipMemset(buf, 0x5a, bytes));
    if (run<64, 8>(buf, bytes, sink)) return 1;
    if (run<48, 8>(buf, bytes, sink)) return 1;
    if (run<32, 8>(buf, bytes, sink)) return 1;
    if (run<64, 4>(buf, bytes, sink)) return 1;
    if (run<48, 4>(buf, bytes, sink)) return 1;
    if (run<64, 16>(buf, bytes, sink)) return 1;
    if (run<48, 16>(buf, bytes, sink)) return 1;
    return 0;
}
