// smem2.hip -- v_sad_u8 throughput vs scalar-operand traffic: REUSE v_sads per loaded SGPR (timing only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int REUSE, bool PIPE>
__global__ void __launch_bounds__(64) kS(const unsigned* __restrict__ tab, unsigned* out, int iters) {
    unsigned acc[REUSE][16];
    unsigned l[REUSE];
#pragma unroll
    for (int r = 0; r < REUSE; ++r) {
        l[r] = threadIdx.x * 2654435761u + r * 77u;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = threadIdx.x + i + r;
    }
    unsigned nxt[16];
    if (PIPE) {
#pragma unroll
        for (int i = 0; i < 16; ++i) nxt[i] = tab[i];
    }
    for (int it = 0; it < iters; ++it) {
        const unsigned* p = tab + (((it + (PIPE ? 1 : 0)) * 16) % 6144);
        unsigned cur[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) cur[i] = PIPE ? nxt[i] : p[i];
        if (PIPE) {
#pragma unroll
            for (int i = 0; i < 16; ++i) nxt[i] = p[i];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int r = 0; r < REUSE; ++r) acc[r][i] = __builtin_amdgcn_sad_u8(l[r], cur[i], acc[r][i]);
    }
    unsigned s = 0;
#pragma unroll
    for (int r = 0; r < REUSE; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[r][i];
    if (s == 0x1234567u) out[0] = s;
}

template <typename F>
static float timeit(F launch, int iters = 10) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

int main() {
    unsigned *tab, *out;
    CHECK(hipMalloc(&tab, 64 * 6144 * 4 + 256)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(tab, 0x21, 64 * 6144 * 4 + 256));
    const int iters = 4000;
#define RUN(REUSE, PIPE, WPS) { \
        const int blocks = 256 * 4 * WPS; \
        const float ms = timeit([&] { kS<REUSE, PIPE><<<blocks, 64>>>(tab, out, iters / REUSE); }); \
        const double ns = ms * 1e6 / ((double)(iters / REUSE) * 16 * REUSE * WPS); \
        printf("reuse=%d pipelined=%d waves/SIMD=%d : %.3f ns per v_sad per SIMD (%.2f cyc @2.4GHz)\n", REUSE, PIPE, WPS, ns, ns * 2.4); }
    RUN(1, false, 4) RUN(1, true, 4) RUN(2, false, 4) RUN(2, true, 4) RUN(4, false, 4) RUN(4, true, 4)
    RUN(1, false, 7) RUN(1, true, 7) RUN(2, false, 7) RUN(2, true, 7) RUN(4, false, 7)
    RUN(1, true, 2) RUN(2, true, 2) RUN(4, true, 2) RUN(2, true, 3) RUN(4, true, 3)
    return 0;
}
