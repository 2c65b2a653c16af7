// sad_variants.hip -- timing-only microbenchmark used to choose the structure of k_sad_tiles.
// Not part of the product library.  Build: hipcc --offload-arch=gfx950 -O3 -o sad_variants sad_variants.hip
// Workload: BASELINE configs[1] (64x64 sensor, 50k views, 16 headings, 3 planes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

constexpr int NPL = 3, APAD = 16;

// ---- A: product structure (waves split q; scalar patch operand; register ring PF deep)
template <int PF, bool SCALAR, bool GLOBAL>
__global__ void __launch_bounds__(256)
kA(const uint4* __restrict__ tiles, const unsigned* __restrict__ prep, unsigned* __restrict__ out, int Q) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nw = blockDim.x >> 6;
    const long long g = blockIdx.x;
    const int q0 = wave * Q / nw, q1 = (wave + 1) * Q / nw;
    const uint4* base = tiles + g * (long long)NPL * Q * 64 + lane;
    unsigned acc[2][APAD];
#pragma unroll
    for (int a = 0; a < APAD; ++a) acc[0][a] = acc[1][a] = 0;
    uint4 ring[PF + 1][NPL];
#pragma unroll
    for (int s = 0; s < PF; ++s)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) ring[s][pl] = GLOBAL ? base[(long long)(pl * Q + min(q0 + s, q1 - 1)) * 64] : make_uint4(lane, s, pl, 1);
    for (int q = q0; q < q1; q += PF + 1) {
#pragma unroll
        for (int s = 0; s <= PF; ++s) {
            const int qc = q + s;
            const int qn = (qc + PF < q1) ? qc + PF : q1 - 1;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) {
                if (GLOBAL) ring[(s + PF) % (PF + 1)][pl] = base[(long long)(pl * Q + qn) * 64];
                else ring[(s + PF) % (PF + 1)][pl] = make_uint4(ring[s][pl].y, ring[s][pl].x + qn, lane, pl);
            }
            if (qc < q1) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const unsigned* pp = prep + (SCALAR ? ((long long)(pl * Q + qc) * 4) * APAD : 0);
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
    unsigned r = 0;
#pragma unroll
    for (int a = 0; a < APAD; ++a) r += acc[0][a] * 3 + acc[1][a];
    atomicAdd(&out[(g * 64 + lane) & 0xffff], r);
}

// ---- D: lockstep structure: the 4 waves of a WG take 4 different view groups and the SAME pixel chunk
// (grid.y chunks), so they read the same patch dwords at the same time (scalar-cache sharing).
template <int PF>
__global__ void __launch_bounds__(256)
kD(const uint4* __restrict__ tiles, const unsigned* __restrict__ prep, unsigned* __restrict__ out, int Q, int G) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long g = (long long)blockIdx.x * 4 + wave;
    if (g >= G) g = G - 1;
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
        for (int pl = 0; pl < NPL; ++pl) ring[s][pl] = base[(long long)(pl * Q + min(q0 + s, q1 - 1)) * 64];
    for (int q = q0; q < q1; q += PF + 1) {
#pragma unroll
        for (int s = 0; s <= PF; ++s) {
            const int qc = q + s;
            const int qn = (qc + PF < q1) ? qc + PF : q1 - 1;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) ring[(s + PF) % (PF + 1)][pl] = base[(long long)(pl * Q + qn) * 64];
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
    unsigned r = 0;
#pragma unroll
    for (int a = 0; a < APAD; ++a) r += acc[0][a] * 3 + acc[1][a];
    atomicAdd(&out[(g * 64 + lane) & 0xffff], r);
}

// ---- E: patch chunk staged in LDS once per WG, read back with wave-uniform (broadcast) ds_read_b128.
// WG = 4 waves = 4 view groups, same pixel chunk (grid.y chunks).  LDS image: [pl][q][j][a] dwords.
template <int PF>
__global__ void __launch_bounds__(256)
kE(const uint4* __restrict__ tiles, const unsigned* __restrict__ prep, unsigned* __restrict__ out, int Q, int G) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    long long g = (long long)blockIdx.x * 4 + wave;
    if (g >= G) g = G - 1;
    const int nchunk = gridDim.y;
    const int q0 = blockIdx.y * Q / nchunk, q1 = (blockIdx.y + 1) * Q / nchunk;
    const int nq = q1 - q0;
    // stage: for each plane, nq*4*APAD dwords contiguous in prep starting at (pl*Q+q0)*4*APAD
    for (int pl = 0; pl < NPL; ++pl) {
        const uint4* src = reinterpret_cast<const uint4*>(prep + ((long long)(pl * Q + q0) * 4) * APAD);
        uint4* dst = reinterpret_cast<uint4*>(lds + (long long)pl * nq * 4 * APAD);
        for (int i = threadIdx.x; i < nq * APAD; i += blockDim.x) dst[i] = src[i];
    }
    __syncthreads();
    const uint4* base = tiles + g * (long long)NPL * Q * 64 + lane;
    unsigned acc[2][APAD];
#pragma unroll
    for (int a = 0; a < APAD; ++a) acc[0][a] = acc[1][a] = 0;
    uint4 ring[PF + 1][NPL];
#pragma unroll
    for (int s = 0; s < PF; ++s)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) ring[s][pl] = base[(long long)(pl * Q + min(q0 + s, q1 - 1)) * 64];
    for (int q = q0; q < q1; q += PF + 1) {
#pragma unroll
        for (int s = 0; s <= PF; ++s) {
            const int qc = q + s;
            const int qn = (qc + PF < q1) ? qc + PF : q1 - 1;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) ring[(s + PF) % (PF + 1)][pl] = base[(long long)(pl * Q + qn) * 64];
            if (qc < q1) {
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) {
                    const uint4* pp = reinterpret_cast<const uint4*>(lds + ((long long)(pl * nq + (qc - q0)) * 4) * APAD);
                    const unsigned lw[4] = {ring[s][pl].x, ring[s][pl].y, ring[s][pl].z, ring[s][pl].w};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int a4 = 0; a4 < APAD / 4; ++a4) {
                            const uint4 p = pp[j * (APAD / 4) + a4];
                            acc[pl == 2][a4 * 4 + 0] = __builtin_amdgcn_sad_u8(lw[j], p.x, acc[pl == 2][a4 * 4 + 0]);
                            acc[pl == 2][a4 * 4 + 1] = __builtin_amdgcn_sad_u8(lw[j], p.y, acc[pl == 2][a4 * 4 + 1]);
                            acc[pl == 2][a4 * 4 + 2] = __builtin_amdgcn_sad_u8(lw[j], p.z, acc[pl == 2][a4 * 4 + 2]);
                            acc[pl == 2][a4 * 4 + 3] = __builtin_amdgcn_sad_u8(lw[j], p.w, acc[pl == 2][a4 * 4 + 3]);
                        }
                }
            }
        }
    }
    unsigned r = 0;
#pragma unroll
    for (int a = 0; a < APAD; ++a) r += acc[0][a] * 3 + acc[1][a];
    atomicAdd(&out[(g * 64 + lane) & 0xffff], r);
}

// ---- G: pure streaming read of the tiles (ceiling)
__global__ void kG(const uint4* __restrict__ src, long long n16, unsigned* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned acc = 0;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        acc += a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    }
    for (; i < n16; i += stride) { const uint4 a = src[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345u) out[0] = acc;
}

template <typename F>
static void timeit(const char* name, double bytes, F launch, int iters = 20) {
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
    ms /= iters;
    printf("%-44s %8.1f us  %7.1f GB/s  (%.1f%% of 8 TB/s)\n", name, ms * 1e3, bytes / ms / 1e6, bytes / ms / 1e6 / 80.0);
}

int main(int argc, char** argv) {
    const int F = argc > 1 ? atoi(argv[1]) : 50000, P = 4096, Q = P / 16;
    const int G = (F + 63) / 64;
    const size_t n16 = (size_t)G * NPL * Q * 64;
    const double bytes = (double)n16 * 16;
    uint4* tiles; unsigned *prep, *out;
    CHECK(hipMalloc(&tiles, n16 * 16));
    CHECK(hipMalloc(&prep, (size_t)NPL * Q * 4 * APAD * 4));
    CHECK(hipMalloc(&out, 65536 * 4));
    std::vector<unsigned> h(n16 * 4);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    CHECK(hipMemcpy(tiles, h.data(), n16 * 16, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(prep, h.data(), (size_t)NPL * Q * 4 * APAD * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(out, 0, 65536 * 4));
    printf("F=%d G=%d bytes=%.1f MB\n", F, G, bytes / 1e6);

    timeit("G  stream read 2048x256", bytes, [&] { kG<<<2048, 256>>>(tiles, (long long)n16, out); });
    timeit("G  stream read 4096x256", bytes, [&] { kG<<<4096, 256>>>(tiles, (long long)n16, out); });
    timeit("A  product PF=2 (scalar+global)", bytes, [&] { kA<2, true, true><<<G, 256>>>(tiles, prep, out, Q); });
    timeit("A  PF=1", bytes, [&] { kA<1, true, true><<<G, 256>>>(tiles, prep, out, Q); });
    timeit("A  PF=3", bytes, [&] { kA<3, true, true><<<G, 256>>>(tiles, prep, out, Q); });
    timeit("A  no scalar loads (global+VALU only)", bytes, [&] { kA<2, false, true><<<G, 256>>>(tiles, prep, out, Q); });
    timeit("A  no global loads (scalar+VALU only)", bytes, [&] { kA<2, true, false><<<G, 256>>>(tiles, prep, out, Q); });
    timeit("A  neither (VALU only)", bytes, [&] { kA<2, false, false><<<G, 256>>>(tiles, prep, out, Q); });
    for (int nchunk : {2, 4, 8, 16}) {
        char nm[64]; snprintf(nm, sizeof nm, "D  lockstep waves, %d pixel chunks", nchunk);
        timeit(nm, bytes, [&] { kD<2><<<dim3((G + 3) / 4, nchunk), 256>>>(tiles, prep, out, Q, G); });
    }
    for (int nchunk : {4, 8, 16}) {
        char nm[64]; snprintf(nm, sizeof nm, "E  LDS-staged patch, %d pixel chunks", nchunk);
        const size_t lds = (size_t)NPL * ((Q + nchunk - 1) / nchunk + 1) * 4 * APAD * 4;
        timeit(nm, bytes, [&] { kE<2><<<dim3((G + 3) / 4, nchunk), 256, lds>>>(tiles, prep, out, Q, G); });
    }
    return 0;
}
