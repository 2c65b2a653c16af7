// sad_trace.hip -- diagnostic build: per-wave placement (XCC, SE, CU) and start/end stamps of the scoring loop.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
constexpr int NPL = 3, APAD = 16;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 ldnt(const uint4* p) { const v4u t = __builtin_nontemporal_load(reinterpret_cast<const v4u*>(p)); return make_uint4(t.x, t.y, t.z, t.w); }

struct Stamp { unsigned long long t0, t1; unsigned hwid, xcc; };

template <int NWG, int PF>
__global__ void __launch_bounds__(64 * NWG)
kT(const uint4* __restrict__ tiles, const unsigned* __restrict__ prep, unsigned* __restrict__ part, int Q, int G, long long Fpad, Stamp* stamps) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long g = (long long)blockIdx.x * NWG + wave;
    const int nchunk = gridDim.y;
    const int q0 = blockIdx.y * Q / nchunk, q1 = (blockIdx.y + 1) * Q / nchunk;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
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
        unsigned* dst = part + (((long long)blockIdx.y * 2) * APAD) * Fpad + g * 64 + lane;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int a = 0; a < APAD; ++a) dst[((long long)s * APAD + a) * Fpad] = acc[s][a];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        const long long w = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * NWG + wave;
        stamps[w] = Stamp{t0, t1, hwid, xcc};
    }
}

template <int NWG, int PF>
static void run(const char* name, int nchunk, const uint4* tiles, const unsigned* prep, unsigned* part, int Q, int G, long long Fpad) {
    const int gx = (G + NWG - 1) / NWG;
    const long long nw = (long long)gx * nchunk * NWG;
    Stamp* d; CHECK(hipMalloc(&d, nw * sizeof(Stamp)));
    for (int i = 0; i < 3; ++i) kT<NWG, PF><<<dim3(gx, nchunk), 64 * NWG>>>(tiles, prep, part, Q, G, Fpad, d);
    CHECK(hipDeviceSynchronize());
    std::vector<Stamp> h(nw);
    CHECK(hipMemcpy(h.data(), d, nw * sizeof(Stamp), hipMemcpyDeviceToHost));
    unsigned long long tmin = ~0ull, tmax = 0;
    for (auto& s : h) { tmin = std::min(tmin, s.t0); tmax = std::max(tmax, s.t1); }
    // per (xcc, se, cu) aggregates.  HW_ID: wave_id[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13] ...
    std::map<unsigned, std::vector<const Stamp*>> percu;
    for (auto& s : h) {
        const unsigned cu = (s.hwid >> 8) & 0xf, sh = (s.hwid >> 12) & 1, se = (s.hwid >> 13) & 7;
        percu[((s.xcc & 0xf) << 12) | (se << 8) | (sh << 4) | cu].push_back(&s);
    }
    std::vector<int> counts; std::vector<double> ends;
    for (auto& kv : percu) {
        counts.push_back((int)kv.second.size());
        unsigned long long e = 0; for (auto* s : kv.second) e = std::max(e, s->t1);
        ends.push_back((e - tmin) / 100.0);
    }
    std::sort(counts.begin(), counts.end()); std::sort(ends.begin(), ends.end());
    double dur = 0; std::vector<double> durs, starts;
    for (auto& s : h) { durs.push_back((s.t1 - s.t0) / 100.0); starts.push_back((s.t0 - tmin) / 100.0); }
    std::sort(durs.begin(), durs.end()); std::sort(starts.begin(), starts.end());
    printf("%-28s waves %6lld span %.1f us | CUs seen %zu waves/CU min %d med %d max %d | CU end us min %.1f med %.1f max %.1f | wave dur us p5 %.1f p50 %.1f p95 %.1f | start p50 %.1f p95 %.1f max %.1f\n",
           name, nw, (tmax - tmin) / 100.0, percu.size(), counts.front(), counts[counts.size() / 2], counts.back(),
           ends.front(), ends[ends.size() / 2], ends.back(), durs[durs.size() / 20], durs[durs.size() / 2], durs[durs.size() * 19 / 20],
           starts[starts.size() / 2], starts[starts.size() * 19 / 20], starts.back());
    // per-XCC end times
    std::map<unsigned, std::pair<int, unsigned long long>> perx;
    for (auto& s : h) { auto& p = perx[s.xcc & 0xf]; p.first++; p.second = std::max(p.second, s.t1); }
    printf("   per XCC (waves,end us):");
    for (auto& kv : perx) printf(" [%u: %d, %.1f]", kv.first, kv.second.first, (kv.second.second - tmin) / 100.0);
    printf("\n");
    CHECK(hipFree(d));
}

int main() {
    const int F = 50000, P = 4096, Q = P / 16, G = (F + 63) / 64;
    const long long Fpad = (long long)G * 64;
    const size_t n16 = (size_t)G * NPL * Q * 64;
    uint4* tiles; unsigned *prep, *part;
    CHECK(hipMalloc(&tiles, n16 * 16));
    CHECK(hipMalloc(&prep, (size_t)NPL * Q * 4 * APAD * 4));
    CHECK(hipMalloc(&part, (size_t)32 * 2 * APAD * Fpad * 4));
    CHECK(hipMemset(tiles, 0x37, n16 * 16));
    CHECK(hipMemset(prep, 0x11, (size_t)NPL * Q * 4 * APAD * 4));
    run<4, 1>("4 g/WG, 8 chunks", 8, tiles, prep, part, Q, G, Fpad);
    run<1, 1>("1 g/WG, 8 chunks", 8, tiles, prep, part, Q, G, Fpad);
    run<4, 1>("4 g/WG, 13 chunks", 13, tiles, prep, part, Q, G, Fpad);
    run<2, 1>("2 g/WG, 8 chunks", 8, tiles, prep, part, Q, G, Fpad);
    run<4, 1>("4 g/WG, 4 chunks", 4, tiles, prep, part, Q, G, Fpad);
    run<4, 2>("4 g/WG, 8 chunks PF2", 8, tiles, prep, part, Q, G, Fpad);
    return 0;
}
