// Probe: where global_load_lds_dwordx3 puts each lane's 12 bytes in LDS (gfx950).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k(const unsigned* src, unsigned* out) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned base = (unsigned)(unsigned long long)(lds_ptr_t)lds;
    const unsigned* g = src + threadIdx.x * 3;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)"
                 : "=&s"(keep) : "v"(g), "s"(base) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(192);
    for (int l = 0; l < 64; ++l) for (int d = 0; d < 3; ++d) h[l * 3 + d] = l * 16 + d;
    unsigned *ds, *dout;
    hipMalloc(&ds, 192 * 4); hipMalloc(&dout, 512 * 4);
    hipMemcpy(ds, h.data(), 192 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, ds, dout);
    std::vector<unsigned> o(512);
    hipMemcpy(o.data(), dout, 512 * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < 272; ++i) { printf("%4x", o[i] == 0xdeadbeefu ? 0xfff : o[i]); if (i % 16 == 15) printf("\n"); }
    return 0;
}
