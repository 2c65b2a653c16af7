// Probe: v_mfma_f32_32x32x64_f8f6f4 with fp4 (E2M1) operands on gfx950 -- exactness of {0,0.5,1,2} x {0,+-1} products under
// the "same K slot in A and B" pairing, and issue rate against v_mfma_i32_32x32x32_i8.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>
typedef int v8i_t __attribute__((ext_vector_type(8)));
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef float v16f_t __attribute__((ext_vector_type(16)));
typedef int v16i_t __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int SCALE>
__global__ void k_one(const uint4* a, const uint4* b, float* d) {
    const int lane = threadIdx.x;
    const uint4 av = a[lane], bv = b[lane];
    v8i_t A = {(int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0};
    v8i_t B = {(int)bv.x, (int)bv.y, (int)bv.z, (int)bv.w, 0, 0, 0, 0};
    v16f_t c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c, 4, 4, 0, SCALE, 0, SCALE);
    for (int r = 0; r < 16; ++r) d[r * 64 + lane] = c[r];
}

// MFMA with its B operand made by VALU right before it (NV v_and per operand dword set), as the scoring kernel does
template <int NV, bool FP4>
__global__ void __launch_bounds__(256) k_rate_valu(const uint4* a, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    const uint4 av = a[lane];
    v8i_t A = {(int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0};
    v4i_t A4 = {(int)av.x, (int)av.y, (int)av.z, (int)av.w};
    v16f_t c[4]; v16i_t ci[4];
    for (int s = 0; s < 4; ++s) for (int r = 0; r < 16; ++r) { c[s][r] = 0.f; ci[s][r] = 0; }
    unsigned x0 = av.x, x1 = av.y, x2 = av.z, x3 = av.w;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            unsigned b0 = x0, b1 = x1, b2 = x2, b3 = x3;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const unsigned m = (0x11111111u << ((s + v) & 3)) | (unsigned)it;
                b0 = (b0 >> v) & m; b1 = (b1 >> v) & m; b2 = (b2 >> v) & m; b3 = (b3 >> v) & m;
            }
            if (FP4) { v8i_t B = {(int)b0, (int)b1, (int)b2, (int)b3, 0, 0, 0, 0}; c[s] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, B, c[s], 4, 4, 0, 0, 0, 0); }
            else { v4i_t B = {(int)b0, (int)b1, (int)b2, (int)b3}; ci[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A4, B, ci[s], 0, 0, 0); }
        }
        x0 += 0x9E3779B9u; x1 ^= x0; x2 += x1; x3 ^= x2;
    }
    float sum = 0.f;
    for (int s = 0; s < 4; ++s) for (int r = 0; r < 16; ++r) sum += c[s][r] + (float)ci[s][r];
    if (sum == 12345.678f) out[0] = sum;
}

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(const uint4* a, float* out, int iters) {
    const int lane = threadIdx.x & 63;
    const uint4 av = a[lane];
    v8i_t A = {(int)av.x, (int)av.y, (int)av.z, (int)av.w, 0, 0, 0, 0};
    v4i_t A4 = {(int)av.x, (int)av.y, (int)av.z, (int)av.w};
    v16f_t c[4];
    v16i_t ci[4];
    for (int s = 0; s < 4; ++s) for (int r = 0; r < 16; ++r) { c[s][r] = 0.f; ci[s][r] = 0; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (MODE == 0) c[s] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A, A, c[s], 4, 4, 0, 0, 0, 0);
            else ci[s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A4, A4, ci[s], 0, 0, 0);
        }
    }
    float sum = 0.f;
    for (int s = 0; s < 4; ++s) for (int r = 0; r < 16; ++r) sum += c[s][r] + (float)ci[s][r];
    if (sum == 12345.678f) out[0] = sum;
}

static float fp4_value(unsigned n) {
    static const float mag[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, 3.f, 4.f, 6.f};
    const float v = mag[n & 7];
    return (n & 8) ? -v : v;
}

int main() {
    std::vector<uint32_t> ha(64 * 4), hb(64 * 4);
    srand(7);
    // A nibbles: 0, +1 (0x2), -1 (0xA); B nibbles: a single bit of {1, 2, 4} (0.5, 1, 2) or 0
    for (int i = 0; i < 256; ++i) {
        uint32_t wa = 0, wb = 0;
        for (int n = 0; n < 8; ++n) {
            const int ra = rand() % 3;
            wa |= (uint32_t)(ra == 0 ? 0x0 : ra == 1 ? 0x2 : 0xA) << (4 * n);
            const int rb = rand() % 4;
            wb |= (uint32_t)(rb == 3 ? 0 : (1u << rb)) << (4 * n);
        }
        ha[i] = wa; hb[i] = wb;
    }
    uint4 *da, *db; float* dd;
    CK(hipMalloc(&da, 1024)); CK(hipMalloc(&db, 1024)); CK(hipMalloc(&dd, 16 * 64 * 4));
    CK(hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice));
    for (int variant = 0; variant < 2; ++variant) {
        if (variant == 0) hipLaunchKernelGGL(k_one<0>, dim3(1), dim3(64), 0, 0, da, db, dd);
        else hipLaunchKernelGGL(k_one<0x7f7f7f7f>, dim3(1), dim3(64), 0, 0, da, db, dd);
        CK(hipDeviceSynchronize());
        std::vector<float> hd(16 * 64);
        CK(hipMemcpy(hd.data(), dd, 16 * 64 * 4, hipMemcpyDeviceToHost));
        // reference: D[row][col] = sum over (half, dword j, nibble n) A[lane = row + 32 half] * B[lane = col + 32 half]
        int bad = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 16; ++r) {
                const int col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float ref = 0.f;
                for (int h = 0; h < 2; ++h)
                    for (int j = 0; j < 4; ++j)
                        for (int n = 0; n < 8; ++n)
                            ref += fp4_value((ha[(row + 32 * h) * 4 + j] >> (4 * n)) & 15) * fp4_value((hb[(col + 32 * h) * 4 + j] >> (4 * n)) & 15);
                if (ref != hd[r * 64 + lane]) { if (bad < 5) printf("  mismatch lane %d r %d: got %g want %g\n", lane, r, hd[r * 64 + lane], ref); ++bad; }
            }
        printf("fp4 exactness (scale %s): %d mismatches of 1024\n", variant ? "0x7f" : "0", bad);
    }
    float* dout; CK(hipMalloc(&dout, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            CK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(k_rate<0>, dim3(256 * 2), dim3(256), 0, 0, da, dout, iters);
            else hipLaunchKernelGGL(k_rate<1>, dim3(256 * 2), dim3(256), 0, 0, da, dout, iters);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double n = 512.0 * 4 * iters * 4;       // MFMA instructions in total
            const double ops = n * 2.0 * 32 * 32 * (mode == 0 ? 64 : 32);
            if (rep) printf("%s: %.3f ms, %.1f Top/s, %.1f ns per MFMA per SIMD\n", mode == 0 ? "fp4 32x32x64" : "i8 32x32x32", ms, ops / ms * 1e-9,
                            ms * 1e6 / (n / 1024.0));
        }
    }
    {
        auto time = [&](auto kern, const char* name) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(kern, dim3(256 * 2), dim3(256), 0, 0, da, dout, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double n = 512.0 * 4 * iters * 4;
                if (rep) printf("%s: %.3f ms, %.1f ns per MFMA per SIMD\n", name, ms, ms * 1e6 / (n / 1024.0));
            }
        };
        time(k_rate_valu<1, true>, "fp4 + 8 VALU per MFMA (1 shift+and per dword)");
        time(k_rate_valu<2, true>, "fp4 + 16 VALU per MFMA");
        time(k_rate_valu<3, true>, "fp4 + 24 VALU per MFMA");
        time(k_rate_valu<1, false>, "i8 + 8 VALU per MFMA");
        time(k_rate_valu<2, false>, "i8 + 16 VALU per MFMA");
    }
    return 0;
}
