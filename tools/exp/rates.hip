// rates.hip -- instruction-rate and streaming-read microbenchmarks (timing only, not product code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

enum { OP_SAD_V = 0, OP_SAD_S, OP_ADD, OP_FMA, OP_DOT4, OP_SAD16, OP_MSAD, OP_PKADD };

template <int OP>
__global__ void kRate(unsigned* out, const unsigned* __restrict__ src, int iters) {
    unsigned acc[16];
    const unsigned l = src[threadIdx.x & 63];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = threadIdx.x + i;
    const unsigned s0 = src[64], s1 = src[65];   // uniform -> SGPR
    float facc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) facc[i] = (float)i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (OP == OP_SAD_V) acc[i] = __builtin_amdgcn_sad_u8(l, acc[(i + 1) & 15] , acc[i]);
                if (OP == OP_SAD_S) acc[i] = __builtin_amdgcn_sad_u8(l, (i & 1) ? s0 : s1, acc[i]);
                if (OP == OP_ADD) acc[i] = acc[i] + (l ^ ((i & 1) ? s0 : s1));
                if (OP == OP_FMA) facc[i] = __builtin_fmaf(facc[i], 1.0001f, (float)l);
                if (OP == OP_DOT4) acc[i] = __builtin_amdgcn_udot4(l, (i & 1) ? s0 : s1, acc[i], false);
                if (OP == OP_SAD16) acc[i] = __builtin_amdgcn_sad_u16(l, (i & 1) ? s0 : s1, acc[i]);
                if (OP == OP_MSAD) acc[i] = __builtin_amdgcn_msad_u8(l, (i & 1) ? s0 : s1, acc[i]);
            }
    }
    unsigned r = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) r += acc[i] + (unsigned)facc[i];
    if (r == 0x1234567u) out[0] = r;
}

template <int UNROLL, bool NT>
__global__ void kRead(const uint4* __restrict__ src, long long n16, unsigned* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    unsigned acc = 0;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n16; i += UNROLL * stride) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if (NT) {
                typedef unsigned v4u __attribute__((ext_vector_type(4)));
                const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(&src[i + u * stride]));
                v[u] = make_uint4(t.x, t.y, t.z, t.w);
            } else v[u] = src[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const uint4 a = src[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345u) out[0] = acc;
}

// contiguous chunk per block (each block streams its own contiguous range)
template <int UNROLL>
__global__ void kReadChunk(const uint4* __restrict__ src, long long n16, unsigned* __restrict__ out) {
    const long long per = (n16 + gridDim.x - 1) / gridDim.x;
    const long long b0 = per * blockIdx.x, b1 = min(n16, b0 + per);
    unsigned acc = 0;
    long long i = b0 + threadIdx.x;
    for (; i + (UNROLL - 1) * blockDim.x < b1; i += UNROLL * blockDim.x) {
        uint4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < b1; i += blockDim.x) { const uint4 a = src[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345u) out[0] = acc;
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
    unsigned *out, *src;
    CHECK(hipMalloc(&out, 4096)); CHECK(hipMalloc(&src, 4096));
    CHECK(hipMemset(src, 0x11, 4096));
    const char* names[] = {"v_sad_u8 vgpr", "v_sad_u8 sgpr", "v_add+xor", "v_fma_f32", "v_dot4_u32_u8", "v_sad_u16", "v_msad_u8"};
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        // grid: 256 CUs x (wps*4 waves) as blocks of 256 threads
        const int blocks = 256 * wps;
        float ms[7];
        ms[0] = timeit([&] { kRate<OP_SAD_V><<<blocks, 256>>>(out, src, iters); });
        ms[1] = timeit([&] { kRate<OP_SAD_S><<<blocks, 256>>>(out, src, iters); });
        ms[2] = timeit([&] { kRate<OP_ADD><<<blocks, 256>>>(out, src, iters); });
        ms[3] = timeit([&] { kRate<OP_FMA><<<blocks, 256>>>(out, src, iters); });
        ms[4] = timeit([&] { kRate<OP_DOT4><<<blocks, 256>>>(out, src, iters); });
        ms[5] = timeit([&] { kRate<OP_SAD16><<<blocks, 256>>>(out, src, iters); });
        ms[6] = timeit([&] { kRate<OP_MSAD><<<blocks, 256>>>(out, src, iters); });
        for (int k = 0; k < 7; ++k) {
            const double inst_per_simd = (double)iters * 64 * wps;   // per wave 64 ops/iter; wps waves per SIMD
            const double ns_per_inst = ms[k] * 1e6 / inst_per_simd;
            printf("waves/SIMD=%d %-16s %8.3f ms  %.3f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n", wps, names[k], ms[k],
                   ns_per_inst, ns_per_inst * 2.4);
        }
    }
    // streaming read ceilings
    for (size_t mb : {615, 2460, 8192}) {
        const size_t n16 = mb * 1000000ull / 16;
        uint4* buf; CHECK(hipMalloc(&buf, n16 * 16)); CHECK(hipMemset(buf, 0x5a, n16 * 16));
        auto rep = [&](const char* nm, float ms) { printf("read %5zu MB %-34s %8.1f us %7.1f GB/s\n", mb, nm, ms * 1e3, n16 * 16 / ms / 1e6); };
        rep("gridstride 2048x256 u4", timeit([&] { kRead<4, false><<<2048, 256>>>(buf, (long long)n16, out); }));
        rep("gridstride 2048x256 u8", timeit([&] { kRead<8, false><<<2048, 256>>>(buf, (long long)n16, out); }));
        rep("gridstride 1024x512 u4", timeit([&] { kRead<4, false><<<1024, 512>>>(buf, (long long)n16, out); }));
        rep("gridstride 4096x256 u2", timeit([&] { kRead<2, false><<<4096, 256>>>(buf, (long long)n16, out); }));
        rep("gridstride 8192x256 u2", timeit([&] { kRead<2, false><<<8192, 256>>>(buf, (long long)n16, out); }));
        rep("gridstride 2048x256 u4 nontemporal", timeit([&] { kRead<4, true><<<2048, 256>>>(buf, (long long)n16, out); }));
        rep("gridstride 2048x256 u8 nontemporal", timeit([&] { kRead<8, true><<<2048, 256>>>(buf, (long long)n16, out); }));
        rep("chunk/block 2048x256 u4", timeit([&] { kReadChunk<4><<<2048, 256>>>(buf, (long long)n16, out); }));
        rep("chunk/block 782x256 u4", timeit([&] { kReadChunk<4><<<782, 256>>>(buf, (long long)n16, out); }));
        rep("chunk/block 3128x256 u4", timeit([&] { kReadChunk<4><<<3128, 256>>>(buf, (long long)n16, out); }));
        rep("chunk/block 12512x256 u4", timeit([&] { kReadChunk<4><<<12512, 256>>>(buf, (long long)n16, out); }));
        CHECK(hipFree(buf));
    }
    return 0;
}
