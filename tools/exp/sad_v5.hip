// sad_v5.hip -- fifth study (timing only): ablations of the balanced single-wave item kernel.
//   MODE 0 full | 1 loads only (xor) | 2 loads + s_loads, 1 VALU op per dword | 3 VALU + s_loads, no vector loads
//   4 full but v_sad replaced by v_xor+v_add (full-rate ops) to see the VALU-rate effect
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int NPL = 3, APAD = 16;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ldnt(const uint4* p) { const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p)); return make_uint4(t.x, t.y, t.z, t.w); }

template <int MODE, int PF, int AEFF>
__global__ void __launch_bounds__(64)
kV(const uint4* __restrict__ tiles, const unsigned* __restrict__ prep, unsigned* __restrict__ part, int Q, int G, long long Fpad, int nchunk) {
    const int lane = threadIdx.x;
    const long long n_items = (long long)G * nchunk;
    for (long long item = blockIdx.x; item < n_items; item += gridDim.x) {
        const int ch = (int)(item / G);
        const long long g = item - (long long)ch * G;
        const int q0 = ch * Q / nchunk, q1 = (ch + 1) * Q / nchunk;
        const uint4* base = tiles + g * (long long)NPL * Q * 64 + lane;
        unsigned acc[2][APAD];
#pragma unroll
        for (int a = 0; a < APAD; ++a) acc[0][a] = acc[1][a] = 0;
        uint4 ring[PF + 1][NPL];
#pragma unroll
        for (int s = 0; s < PF; ++s)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) ring[s][pl] = MODE == 3 ? make_uint4(lane, s, pl, 7) : ldnt(&base[(long long)(pl * Q + min(q0 + s, q1 - 1)) * 64]);
        for (int q = q0; q < q1; q += PF + 1) {
#pragma unroll
            for (int s = 0; s <= PF; ++s) {
                const int qc = q + s;
                const int qn = (qc + PF < q1) ? qc + PF : q1 - 1;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    if (MODE == 3) ring[(s + PF) % (PF + 1)][pl] = make_uint4(ring[s][pl].y + qn, ring[s][pl].z, ring[s][pl].w, ring[s][pl].x);
                    else ring[(s + PF) % (PF + 1)][pl] = ldnt(&base[(long long)(pl * Q + qn) * 64]);
                }
                if (qc < q1) {
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) {
                        const unsigned* pp = prep + ((long long)(pl * Q + qc) * 4) * APAD;
                        const unsigned lw[4] = {ring[s][pl].x, ring[s][pl].y, ring[s][pl].z, ring[s][pl].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (MODE == 1) acc[pl == 2][j] ^= lw[j];
                            else if (MODE == 2) acc[pl == 2][j] += lw[j] ^ pp[j * APAD] ^ pp[j * APAD + 15];
                            else {
#pragma unroll
                                for (int a = 0; a < AEFF; ++a) {
                                    if (MODE == 4) acc[pl == 2][a] += lw[j] ^ pp[j * APAD + a];
                                    else acc[pl == 2][a] = __builtin_amdgcn_sad_u8(lw[j], pp[j * APAD + a], acc[pl == 2][a]);
                                }
                            }
                        }
                    }
                }
            }
        }
        unsigned* dst = part + (((long long)ch * 2) * APAD) * Fpad + g * 64 + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int a = 0; a < APAD; ++a) dst[((long long)s * APAD + a) * Fpad] = acc[s][a];
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
#define RUN(MODE, PF, AEFF, NCH, WPC, label) { \
        const long long items = (long long)G * NCH; \
        const long long grid = items < 256ll * WPC ? items : 256ll * WPC; \
        float best = 1e9; for (int r = 0; r < 3; ++r) { const float ms = timeit([&] { kV<MODE, PF, AEFF><<<dim3((unsigned)grid), 64>>>(tiles, prep, part, Q, G, Fpad, NCH); }); best = ms < best ? ms : best; } \
        printf("%-44s PF=%d A=%2d chunks=%2d grid=%5lld : %7.1f us (%.1f%%)\n", label, PF, AEFF, NCH, grid, best * 1e3, bytes / best / 1e6 / 80.0); }
    RUN(0, 1, 16, 7, 28, "full")
    RUN(1, 1, 16, 7, 28, "vector loads only")
    RUN(2, 1, 16, 7, 28, "vector + scalar loads, ~no VALU")
    RUN(3, 1, 16, 7, 28, "VALU + scalar loads, no vector loads")
    RUN(4, 1, 16, 7, 28, "full with full-rate VALU (xor+add)")
    RUN(0, 1, 8, 7, 28, "full, 8 headings of VALU")
    RUN(0, 1, 4, 7, 28, "full, 4 headings of VALU")
    RUN(0, 2, 16, 7, 28, "full PF=2")
    RUN(1, 2, 16, 7, 28, "vector loads only PF=2")
    RUN(1, 3, 16, 7, 28, "vector loads only PF=3")
    RUN(1, 1, 16, 4, 28, "vector loads only, 4 chunks")
    RUN(1, 1, 16, 14, 28, "vector loads only, 14 chunks (2 rounds)")
    RUN(0, 1, 16, 8, 28, "full, 8 chunks")
    RUN(0, 1, 16, 9, 28, "full, 9 chunks")
    return 0;
}
