// sad_v4.hip -- fourth study (timing only): what bounds a wave -- scalar-load latency, vector-load latency, VALU?
// Variants of the single-wave item kernel: HOT patch (all s_loads hit the scalar cache), VL=2 (two view groups per
// lane -> twice the VALU work per s_load), PF depth, waves/SIMD cap.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int NPL = 3, APAD = 16;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ldnt(const uint4* p) { const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p)); return make_uint4(t.x, t.y, t.z, t.w); }

template <int VL, int PF, bool HOT>
__global__ void __launch_bounds__(64)
kV(const uint4* __restrict__ tiles, const unsigned* __restrict__ prep, unsigned* __restrict__ part, int Q, int G, long long Fpad, int nchunk) {
    const int lane = threadIdx.x;
    const int GV = (G + VL - 1) / VL;
    const long long n_items = (long long)GV * nchunk;
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ch = (int)(item / GV);
        const long long gv = item - (long long)ch * GV;
        const int q0 = ch * Q / nchunk, q1 = (ch + 1) * Q / nchunk;
        const uint4* base[VL];
#pragma unroll
        for (int v = 0; v < VL; ++v) { long long g = gv * VL + v; if (g >= G) g = G - 1; base[v] = tiles + g * (long long)NPL * Q * 64 + lane; }
        unsigned acc[VL][2][APAD];
#pragma unroll
        for (int v = 0; v < VL; ++v)
#pragma unroll
            for (int a = 0; a < APAD; ++a) acc[v][0][a] = acc[v][1][a] = 0;
        uint4 ring[PF + 1][VL][NPL];
#pragma unroll
        for (int s = 0; s < PF; ++s)
#pragma unroll
            for (int v = 0; v < VL; ++v)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) ring[s][v][pl] = ldnt(&base[v][(long long)(pl * Q + min(q0 + s, q1 - 1)) * 64]);
        for (int q = q0; q < q1; q += PF + 1) {
#pragma unroll
            for (int s = 0; s <= PF; ++s) {
                const int qc = q + s;
                const int qn = (qc + PF < q1) ? qc + PF : q1 - 1;
#pragma unroll
                for (int v = 0; v < VL; ++v)
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) ring[(s + PF) % (PF + 1)][v][pl] = ldnt(&base[v][(long long)(pl * Q + qn) * 64]);
                if (qc < q1) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                        const unsigned* pp = prep + ((long long)(pl * Q + (HOT ? 0 : qc)) * 4) * APAD;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
#pragma unroll
                            for (int a = 0; a < APAD; ++a) {
                                const unsigned p = pp[j * APAD + a];
#pragma unroll
                                for (int v = 0; v < VL; ++v) {
                                    const unsigned lw = j == 0 ? ring[s][v][pl].x : j == 1 ? ring[s][v][pl].y : j == 2 ? ring[s][v][pl].z : ring[s][v][pl].w;
                                    acc[v][pl == 2][a] = __builtin_amdgcn_sad_u8(lw, p, acc[v][pl == 2][a]);
                                }
                            }
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < VL; ++v) {
            const long long g = gv * VL + v;
            if (g < G) {
                unsigned* dst = part + (((long long)ch * 2) * APAD) * Fpad + g * 64 + lane;
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int a = 0; a < APAD; ++a) dst[((long long)s * APAD + a) * Fpad] = acc[v][s][a];
            }
        }
    }
}

template <typename F>
static float timeit(F launch, int iters = 20) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch();
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
    const int F = 50000, P = 4096, Q = P / 16, G = (F + 63) / 64;
    const long long Fpad = (long long)G * 64;
    const size_t n16 = (size_t)G * NPL * Q * 64;
    const double bytes = (double)n16 * 16;
    uint4* tiles; unsigned *prep, *part;
    CHECK(hipMalloc(&tiles, n16 * 16));
    CHECK(hipMalloc(&prep, (size_t)NPL * Q * 4 * APAD * 4));
    CHECK(hipMalloc(&part, (size_t)32 * 2 * APAD * Fpad * 4));
    std::vector<unsigned> h(n16 * 4);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    CHECK(hipMemcpy(tiles, h.data(), n16 * 16, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(prep, h.data(), (size_t)NPL * Q * 4 * APAD * 4, hipMemcpyHostToDevice));
#define RUN(VL, PF, HOT, NCH, WPC) { \
        const long long items = (long long)((G + VL - 1) / VL) * NCH; \
        const long long grid = items < 256ll * WPC ? items : 256ll * WPC; \
        const float ms = timeit([&] { kV<VL, PF, HOT><<<dim3((unsigned)grid), 64>>>(tiles, prep, part, Q, G, Fpad, NCH); }); \
        printf("VL=%d PF=%d HOT=%d chunks=%2d wpc=%2d items=%5lld grid=%5lld : %7.1f us (%.1f%%)\n", VL, PF, HOT, NCH, WPC, items, grid, ms * 1e3, bytes / ms / 1e6 / 80.0); }
    RUN(1, 1, false, 7, 24) RUN(1, 1, true, 7, 24) RUN(1, 2, false, 7, 24) RUN(1, 2, true, 7, 24) RUN(1, 3, true, 7, 24)
    RUN(1, 1, false, 7, 28) RUN(1, 1, true, 7, 28) RUN(1, 1, true, 8, 28) RUN(1, 2, true, 8, 28)
    RUN(2, 1, false, 10, 16) RUN(2, 1, true, 10, 16) RUN(2, 1, false, 12, 20) RUN(2, 1, true, 12, 20) RUN(2, 1, false, 14, 24) RUN(2, 1, true, 14, 24)
    RUN(2, 2, false, 10, 16) RUN(2, 2, true, 10, 16) RUN(2, 1, false, 8, 16) RUN(2, 1, false, 8, 12)
    return 0;
}
