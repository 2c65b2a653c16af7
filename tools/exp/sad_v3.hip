// sad_v3.hip -- third structure study (timing only).  Workgroup = NWG view groups x NWC pixel sub-chunks;
// sub-chunk sums are reduced through LDS (ds_add) before the partial store; grid.y = chunks across workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int NPL = 3, APAD = 16;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ldnt(const uint4* p) { const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p)); return make_uint4(t.x, t.y, t.z, t.w); }

template <int NWG, int NWC, int PF, int ORDER>
__global__ void __launch_bounds__(64 * NWG * NWC)
kT(const uint4* __restrict__ tiles, const unsigned* __restrict__ prep, unsigned* __restrict__ part, int Q, int G, long long Fpad) {
    __shared__ unsigned red[NWC > 1 ? NWG * 2 * APAD * 64 : 1];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = wave % NWG, wc = wave / NWG;
    int bx = blockIdx.x, by = blockIdx.y;
    if (ORDER == 1) {   // g-major: consecutive workgroups = consecutive chunks of the same view groups
        const int lin = blockIdx.y * gridDim.x + blockIdx.x;
        by = lin % gridDim.y; bx = lin / gridDim.y;
    }
    const long long g = (long long)bx * NWG + wg;
    const int nchunk = gridDim.y * NWC;
    const int ch = by * NWC + wc;
    const int q0 = ch * Q / nchunk, q1 = (ch + 1) * Q / nchunk;
    if (NWC > 1) { for (int i = threadIdx.x; i < NWG * 2 * APAD * 64; i += blockDim.x) red[i] = 0; __syncthreads(); }
    unsigned acc[2][APAD];
#pragma unroll
    for (int a = 0; a < APAD; ++a) acc[0][a] = acc[1][a] = 0;
    if (g < G) {
        const uint4* base = tiles + g * (long long)NPL * Q * 64 + lane;
        uint4 ring[PF + 1][NPL];
#pragma unroll
        for (int s = 0; s < PF; ++s)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) ring[s][pl] = ldnt(&base[(long long)(pl * Q + min(q0 + s, q1 - 1)) * 64]);
        for (int q = q0; q < q1; q += PF + 1) {
#pragma unroll
            for (int s = 0; s <= PF; ++s) {
                const int qc = q + s;
                const int qn = (qc + PF < q1) ? qc + PF : q1 - 1;
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) ring[(s + PF) % (PF + 1)][pl] = ldnt(&base[(long long)(pl * Q + qn) * 64]);
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
    }
    if (NWC > 1) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int a = 0; a < APAD; ++a) atomicAdd(&red[((wg * 2 + s) * APAD + a) * 64 + lane], acc[s][a]);
        __syncthreads();
        // NWG*2*APAD rows of 64 values; thread t stores rows t/64, ...
        for (int r = wave; r < NWG * 2 * APAD; r += NWG * NWC) {
            const int rg = r / (2 * APAD), rs = r % (2 * APAD);
            const long long gg = (long long)bx * NWG + rg;
            if (gg < G) part[(((long long)by * 2 * APAD) + rs) * Fpad + gg * 64 + lane] = red[r * 64 + lane];
        }
    } else if (g < G) {
        unsigned* dst = part + (((long long)by * 2) * APAD) * Fpad + g * 64 + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int a = 0; a < APAD; ++a) dst[((long long)s * APAD + a) * Fpad] = acc[s][a];
    }
}

__global__ void kCombine(const unsigned* __restrict__ part, double* __restrict__ fam, unsigned long long* __restrict__ bmax, int nchunk, long long Fpad, long long F) {
    __shared__ unsigned long long wm[4];
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
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x == 0) { unsigned long long m = wm[0]; for (int i = 1; i < 4; ++i) m = wm[i] > m ? wm[i] : m; bmax[(long long)a * gridDim.x + blockIdx.x] = m; }
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
    uint4* tiles; unsigned *prep, *part; double* fam; unsigned long long* bmax;
    CHECK(hipMalloc(&tiles, n16 * 16));
    CHECK(hipMalloc(&prep, (size_t)NPL * Q * 4 * APAD * 4));
    CHECK(hipMalloc(&part, (size_t)32 * 2 * APAD * Fpad * 4));
    CHECK(hipMalloc(&fam, (size_t)APAD * Fpad * 8));
    CHECK(hipMalloc(&bmax, 64 * 1024 * 8));
    std::vector<unsigned> h(n16 * 4);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    CHECK(hipMemcpy(tiles, h.data(), n16 * 16, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(prep, h.data(), (size_t)NPL * Q * 4 * APAD * 4, hipMemcpyHostToDevice));
    printf("F=%d G=%d bytes=%.1f MB\n", F, G, bytes / 1e6);
#define RUN(NWG, NWC, PF, ORDER, GY) { \
        const float ms = timeit([&] { kT<NWG, NWC, PF, ORDER><<<dim3((G + NWG - 1) / NWG, GY), 64 * NWG * NWC>>>(tiles, prep, part, Q, G, Fpad); }); \
        const float mc = timeit([&] { kCombine<<<dim3((unsigned)((Fpad + 255) / 256), APAD), 256>>>(part, fam, bmax, GY, Fpad, F); }); \
        printf("g/WG=%d sub/WG=%d PF=%d order=%d gridchunks=%2d WGs=%5d waves=%5d : main %7.1f us (%.1f%%)  combine %5.1f us  sum %7.1f\n", NWG, NWC, PF, ORDER, GY, \
               ((G + NWG - 1) / NWG) * GY, ((G + NWG - 1) / NWG) * GY * NWG * NWC, ms * 1e3, bytes / ms / 1e6 / 80.0, mc * 1e3, (ms + mc) * 1e3); }
    RUN(4, 1, 1, 0, 8) RUN(4, 1, 1, 1, 8) RUN(4, 1, 1, 0, 7) RUN(4, 1, 1, 0, 6) RUN(4, 1, 1, 0, 10) RUN(4, 1, 1, 0, 12)
    RUN(1, 4, 1, 0, 2) RUN(1, 4, 1, 0, 3) RUN(1, 4, 1, 0, 4) RUN(1, 2, 1, 0, 4) RUN(1, 2, 1, 0, 3)
    RUN(2, 2, 1, 0, 4) RUN(2, 2, 1, 0, 3) RUN(2, 4, 1, 0, 2) RUN(4, 2, 1, 0, 4) RUN(4, 2, 1, 0, 3) RUN(4, 4, 1, 0, 2)
    RUN(1, 8, 1, 0, 1) RUN(2, 4, 1, 0, 1) RUN(1, 4, 2, 0, 2) RUN(2, 2, 2, 0, 4) RUN(8, 1, 1, 0, 8) RUN(2, 1, 1, 0, 8)
    return 0;
}
