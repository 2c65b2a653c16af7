// sad_v2.hip -- second structure study (timing only): items = (view group, pixel chunk), partial sums to
// global memory, separate combine kernel.  Workload: 64x64 sensor, 50k views, 16 headings, 3 planes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int NPL = 3, APAD = 16;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ uint4 ld(const uint4* p) {
    if (NT) { const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p)); return make_uint4(t.x, t.y, t.z, t.w); }
    return *p;
}

// block = 64*NW threads; wave w of block b takes view group b*NW + w; blockIdx.y = pixel chunk.
template <int PF, bool NT>
__global__ void __launch_bounds__(256)
kS(const uint4* __restrict__ tiles, const unsigned* __restrict__ prep, unsigned* __restrict__ part, int Q, int G, long long Fpad) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const long long g = (long long)blockIdx.x * nw + wave;
    if (g >= G) return;
    const int nchunk = gridDim.y;
    const int q0 = blockIdx.y * Q / nchunk, q1 = (blockIdx.y + 1) * Q / nchunk;
    const uint4* base = tiles + g * (long long)NPL * Q * 64 + lane;
    unsigned acc[2][APAD];
#pragma unroll
    for (int a = 0; a < APAD; ++a) acc[0][a] = acc[1][a] = 0;
    uint4 ring[PF + 1][NPL];
#pragma unroll
    for (int s = 0; s < PF; ++s)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) ring[s][pl] = ld<NT>(&base[(long long)(pl * Q + min(q0 + s, q1 - 1)) * 64]);
    for (int q = q0; q < q1; q += PF + 1) {
#pragma unroll
        for (int s = 0; s <= PF; ++s) {
            const int qc = q + s;
            const int qn = (qc + PF < q1) ? qc + PF : q1 - 1;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) ring[(s + PF) % (PF + 1)][pl] = ld<NT>(&base[(long long)(pl * Q + qn) * 64]);
            if (qc < q1) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const unsigned* pp = prep + ((long long)(pl * Q + qc) * 4) * APAD;
                    const unsigned lw[4] = {ring[s][pl].x, ring[s][pl].y, ring[s][pl].z, ring[s][pl].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int a = 0; a < APAD; ++a)
                            acc[pl == 2][a] = __builtin_amdgcn_sad_u8(lw[j], pp[j * APAD + a], acc[pl == 2][a]);
                }
            }
        }
    }
    // partial sums: part[chunk][s][a][f]
    unsigned* dst = part + (((long long)blockIdx.y * 2) * APAD) * Fpad + g * 64 + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int a = 0; a < APAD; ++a) dst[((long long)s * APAD + a) * Fpad] = acc[s][a];
}

__global__ void kCombine(const unsigned* __restrict__ part, double* __restrict__ fam, unsigned long long* amax, int nchunk, long long Fpad, long long F) {
    const long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y;
    unsigned long long key = 0;
    if (f < F) {
        unsigned hs = 0, v = 0;
        for (int c = 0; c < nchunk; ++c) {
            hs += part[(((long long)c * 2 + 0) * APAD + a) * Fpad + f];
            v += part[(((long long)c * 2 + 1) * APAD + a) * Fpad + f];
        }
        const double val = 4096.0 - (0.125 * hs + 0.75 * v) / 255.;
        fam[a * Fpad + f] = val;
        key = (unsigned long long)__double_as_longlong(val) | 0x8000000000000000ull;
    }
    for (int o = 32; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(key, o); key = other > key ? other : key; }
    if ((threadIdx.x & 63) == 0) atomicMax(&amax[a], key);
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

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 50000, P = 4096, Q = P / 16;
    const int G = (F + 63) / 64;
    const long long Fpad = (long long)G * 64;
    const size_t n16 = (size_t)G * NPL * Q * 64;
    const double bytes = (double)n16 * 16;
    uint4* tiles; unsigned *prep, *part; double* fam; unsigned long long* amax;
    CHECK(hipMalloc(&tiles, n16 * 16));
    CHECK(hipMalloc(&prep, (size_t)NPL * Q * 4 * APAD * 4));
    CHECK(hipMalloc(&part, (size_t)16 * 2 * APAD * Fpad * 4));
    CHECK(hipMalloc(&fam, (size_t)APAD * Fpad * 8));
    CHECK(hipMalloc(&amax, 64 * 8));
    std::vector<unsigned> h(n16 * 4);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    CHECK(hipMemcpy(tiles, h.data(), n16 * 16, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(prep, h.data(), (size_t)NPL * Q * 4 * APAD * 4, hipMemcpyHostToDevice));
    printf("F=%d G=%d bytes=%.1f MB\n", F, G, bytes / 1e6);
#define RUN(PF, NT, NW, NCH) { \
        const float ms = timeit([&] { kS<PF, NT><<<dim3((G + NW - 1) / NW, NCH), 64 * NW>>>(tiles, prep, part, Q, G, Fpad); }); \
        const float mc = timeit([&] { kCombine<<<dim3((unsigned)((Fpad + 255) / 256), APAD), 256>>>(part, fam, amax, NCH, Fpad, F); }); \
        printf("PF=%d NT=%d waves/WG=%d chunks=%2d items=%5d : main %7.1f us %7.1f GB/s (%.1f%%)  combine %5.1f us\n", PF, NT, NW, NCH, \
               ((G + NW - 1) / NW) * NCH, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0, mc * 1e3); }
    RUN(2, false, 1, 4) RUN(2, false, 1, 6) RUN(2, false, 1, 8) RUN(2, false, 1, 16)
    RUN(2, false, 2, 4) RUN(2, false, 2, 8)
    RUN(2, false, 4, 4) RUN(2, false, 4, 6) RUN(2, false, 4, 8) RUN(2, false, 4, 16)
    RUN(1, false, 1, 8) RUN(3, false, 1, 8) RUN(1, false, 4, 8) RUN(3, false, 4, 8)
    RUN(2, true, 1, 8) RUN(2, true, 4, 8) RUN(2, true, 4, 6) RUN(1, true, 4, 8) RUN(3, true, 4, 8) RUN(2, true, 1, 16)
    return 0;
}
