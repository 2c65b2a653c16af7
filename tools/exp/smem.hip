// smem.hip -- how do scalar loads and v_sad_u8 share a SIMD?  (timing only)
// Each wave: loop { NL x s_load_dwordx16 from a table (stride per iteration), 16*NL v_sad_u8 with SGPR operands }.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>   // 0: scalar table walks 24 KB window; 1: fixed 64 B (always hit); 2: no scalar loads; 3: per-wave distinct 24 KB windows (L2 misses in K$)
__global__ void __launch_bounds__(64) kS(const unsigned* __restrict__ tab, unsigned* out, int iters, int tabdw) {
    unsigned acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x + i;
    const unsigned l = threadIdx.x * 2654435761u;
    const unsigned* base = tab + (MODE == 3 ? (blockIdx.x % 64) * 6144 : 0);
    unsigned fixed[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) fixed[i] = tab[i];
    for (int it = 0; it < iters; ++it) {
        const unsigned* p = base + (MODE == 1 ? 0 : ((it * 16) % 6144));
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned s = MODE == 2 ? fixed[i] : p[i];
            acc[i] = __builtin_amdgcn_sad_u8(l, s, acc[i]);
        }
    }
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i];
    if (r == 0x1234567u) out[0] = r;
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
    const int tabdw = 64 * 6144 + 64;
    CHECK(hipMalloc(&tab, tabdw * 4)); CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(tab, 0x21, tabdw * 4));
    const int iters = 4000;
    const char* names[] = {"s_load walking 24 KB (shared)", "s_load fixed line", "no s_load (SGPR resident)", "s_load walking per-wave windows"};
    for (int wps : {1, 2, 4, 6, 7, 8}) {
        const int blocks = 256 * 4 * wps;   // single-wave blocks
        float ms[4];
        ms[0] = timeit([&] { kS<0><<<blocks, 64>>>(tab, out, iters, tabdw); });
        ms[1] = timeit([&] { kS<1><<<blocks, 64>>>(tab, out, iters, tabdw); });
        ms[2] = timeit([&] { kS<2><<<blocks, 64>>>(tab, out, iters, tabdw); });
        ms[3] = timeit([&] { kS<3><<<blocks, 64>>>(tab, out, iters, tabdw); });
        for (int k = 0; k < 4; ++k) {
            const double ns = ms[k] * 1e6 / ((double)iters * 16 * wps);
            printf("waves/SIMD=%d %-34s %8.3f ms  %.3f ns per v_sad per SIMD (%.2f cyc @2.4GHz)\n", wps, names[k], ms[k], ns, ns * 2.4);
        }
    }
    return 0;
}
