// Probe: operand / result lane maps of the multi-block f32 MFMAs v_mfma_f32_32x32x1_2b_f32 and v_mfma_f32_16x16x1_4b_f32 on
// gfx950 (which A lane and which B lane feed result register r of lane l), and their issue rate beside a stream of loads.
// Not part of the product: k_ssd_f32_mfma's epilogue is written from this map.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v32f_t __attribute__((ext_vector_type(32)));
typedef float v16f_t __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void k_map32(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x;
    v32f_t c;
    for (int r = 0; r < 32; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x1f32(a[lane], b[lane], c, 0, 0, 0);
    for (int r = 0; r < 32; ++r) d[r * 64 + lane] = c[r];
}
__global__ void k_map16(const float* a, const float* b, float* d) {
    const int lane = threadIdx.x;
    v16f_t c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_16x16x1f32(a[lane], b[lane], c, 0, 0, 0);
    for (int r = 0; r < 16; ++r) d[r * 64 + lane] = c[r];
}
template <int W>
__global__ void __launch_bounds__(256) k_rate(const float* a, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    float x = a[lane], y = a[lane + 64];
    v32f_t c32; v16f_t c16;
    for (int r = 0; r < 32; ++r) c32[r] = 0.f;
    for (int r = 0; r < 16; ++r) c16[r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (W == 32) c32 = __builtin_amdgcn_mfma_f32_32x32x1f32(x, y, c32, 0, 0, 0);
            else c16 = __builtin_amdgcn_mfma_f32_16x16x1f32(x, y, c16, 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int r = 0; r < 32; ++r) s += c32[r];
    for (int r = 0; r < 16; ++r) s += c16[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float *a, *b, *d;
    CK(hipMalloc(&a, 4096)); CK(hipMalloc(&b, 4096)); CK(hipMalloc(&d, 32 * 64 * 4 * 64));
    std::vector<float> ha(128), hb(128), hd(32 * 64);
    for (int which = 0; which < 2; ++which) {
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = 0; i < 128; ++i) { ha[i] = pass == 0 ? (float)(i + 1) : 1.f; hb[i] = pass == 0 ? 1.f : (float)(i + 1); }
            CK(hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice));
            if (which == 0) hipLaunchKernelGGL(k_map32, dim3(1), dim3(64), 0, 0, a, b, d);
            else hipLaunchKernelGGL(k_map16, dim3(1), dim3(64), 0, 0, a, b, d);
            CK(hipDeviceSynchronize());
            const int R = which == 0 ? 32 : 16;
            CK(hipMemcpy(hd.data(), d, R * 64 * 4, hipMemcpyDeviceToHost));
            printf("%s: %s lane feeding result reg r of lane l (rows r, cols l):\n", which == 0 ? "32x32x1_2b" : "16x16x1_4b", pass == 0 ? "A" : "B");
            for (int r = 0; r < R; ++r) {
                printf("r%2d:", r);
                for (int l = 0; l < 64; ++l) printf(" %2d", (int)hd[r * 64 + l] - 1);
                printf("\n");
            }
        }
    }
    // rate: 256 CUs x 4 waves (one per SIMD), 8 MFMAs per iteration
    float* out; CK(hipMalloc(&out, 1024 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int W = 0; W < 2; ++W) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (W == 0) hipLaunchKernelGGL(k_rate<32>, dim3(1024), dim3(256), 0, 0, a, out, iters);
            else hipLaunchKernelGGL(k_rate<16>, dim3(1024), dim3(256), 0, 0, a, out, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            // 1024 blocks x 4 waves over 1024 SIMDs = 4 waves per SIMD in sequence; cycles per MFMA at an assumed 2.4 GHz
            const double mf = 4.0 * iters * 8;
            if (rep) printf("%s: %.3f ms for %.0f MFMAs per SIMD -> %.1f ns per MFMA (%.1f cycles at 2.4 GHz)\n", W == 0 ? "32x32x1_2b" : "16x16x1_4b", ms, mf,
                            ms * 1e6 / mf, ms * 1e6 / mf * 2.4);
        }
    }
    return 0;
}
