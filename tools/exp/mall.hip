// mall.hip -- does the 256 MiB Infinity Cache keep a default-policy-read subset resident while the rest of a
// 615 MB buffer streams past with non-temporal loads?  (timing only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// each block streams a contiguous chunk; chunks below `hot16` use default-policy loads, the rest POLICY loads
template <int POLICY>   // 0 default, 1 nontemporal, 2 sc1 (L2 bypass... agent-scope relaxed atomic-ish load)
__global__ void kRead(const uint4* __restrict__ src, long long n16, long long hot16, unsigned* out) {
    const long long per = (n16 + gridDim.x - 1) / gridDim.x;
    const long long b0 = per * blockIdx.x, b1 = min(n16, b0 + per);
    unsigned acc = 0;
    const bool hot = b0 < hot16;
    for (long long i = b0 + threadIdx.x; i < b1; i += 4 * blockDim.x) {
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long long j = i + u * blockDim.x;
            if (j < b1) {
                if (hot || POLICY == 0) v[u] = src[j];
                else { const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(&src[j])); v[u] = make_uint4(t.x, t.y, t.z, t.w); }
            } else v[u] = make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345u) out[0] = acc;
}

template <typename F>
static float timeit(F launch, int iters = 30) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
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
    unsigned* out; CHECK(hipMalloc(&out, 64));
    for (size_t mb : {615, 205, 410}) {
        const size_t n16 = mb * 1000000ull / 16;
        uint4* buf; CHECK(hipMalloc(&buf, n16 * 16)); CHECK(hipMemset(buf, 0x5a, n16 * 16));
        for (int hotmb : {0, 64, 128, 192, 224, (int)mb}) {
            if (hotmb > (int)mb) continue;
            const long long hot16 = (long long)hotmb * 1000000ll / 16;
            const float t1 = timeit([&] { kRead<1><<<6256, 256>>>(buf, (long long)n16, hot16, out); });
            printf("buffer %4zu MB, default-policy part %4d MB, rest non-temporal : %7.1f us  %7.1f GB/s\n", mb, hotmb, t1 * 1e3, n16 * 16 / t1 / 1e6);
        }
        CHECK(hipFree(buf));
    }
    return 0;
}
