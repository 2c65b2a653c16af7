// graph_chain.hip -- is a hipGraph a faster way to put an agent step's three dependent kernels (preparation ~8 us, scoring
// ~36 us, fold ~7 us) on the device than three stream launches?  The host's time per step is measured from "launch" to "the
// record has arrived in mapped host memory", as dv_agent_step waits.  Three forms:
//   stream        three hipLaunchKernelGGL on one stream, per-step arguments (sequence number, pose) as kernel arguments;
//   graph         the same chain captured once; the per-step arguments are read from a mapped host word the host writes before
//                 hipGraphLaunch (no node is updated);
//   graph+params  hipGraphExecKernelNodeSetParams on the first and the last node before every launch (what changing kernel
//                 arguments costs).
// Kernels: stand-ins of the real ones' grid shapes that spin on the clock for a given time (timing only; nothing is computed).
//   hipcc --offload-arch=gfx950 -O3 -o graph_chain graph_chain.hip && ./graph_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ void spin_us(float us) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();           // 100 MHz
    const unsigned long long ticks = (unsigned long long)(us * 100.f);
    for (int i = 0; i < 100000 && __builtin_amdgcn_s_memrealtime() - t0 < ticks; ++i) __builtin_amdgcn_s_sleep(1);      // (bounded whatever the clock does)
}

__global__ void __launch_bounds__(256) k_prep(const volatile unsigned long long* args, unsigned long long seq_arg, unsigned* scratch, float us) {
    spin_us(us);
    if (threadIdx.x == 0 && blockIdx.x == 0) scratch[0] = (unsigned)(args ? args[0] : seq_arg);     // (one lane reads the mapped word)
}
__global__ void __launch_bounds__(512) k_score(unsigned* scratch, float us) {
    spin_us(us);
    if (threadIdx.x == 0) scratch[1 + blockIdx.x] = scratch[0];
}
__global__ void __launch_bounds__(256) k_fold(const unsigned* scratch, volatile unsigned long long* record, float us) {
    spin_us(us);
    if (threadIdx.x == 0) record[0] = scratch[1];              // the record: the step's sequence number, to mapped host memory
}

int main() {
    hipStream_t st;
    CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    unsigned* scratch;
    CHECK(hipMalloc(&scratch, 4096));
    unsigned long long *h_rec, *d_rec, *h_args, *d_args;
    CHECK(hipHostMalloc(&h_rec, 64, hipHostMallocMapped));
    CHECK(hipHostMalloc(&h_args, 64, hipHostMallocMapped));
    CHECK(hipHostGetDevicePointer((void**)&d_rec, h_rec, 0));
    CHECK(hipHostGetDevicePointer((void**)&d_args, h_args, 0));
    const float t_prep = 5.f, t_score = 33.f, t_fold = 4.f;   // kernel bodies; launch overheads come on top
    volatile unsigned long long* rec = h_rec;
    auto wait = [&](unsigned long long seq) {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        while (*rec != seq) {
            if ((++spins & 0xfffff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(2)) {
                printf("record %llu never arrived (have %llu): %s\n", seq, (unsigned long long)*rec, hipGetErrorString(hipStreamQuery(st)));
                fflush(stdout);
                exit(2);
            }
        }
    };
    unsigned long long seq = 0;
    auto stream_step = [&]() {
        ++seq;
        hipLaunchKernelGGL(k_prep, dim3(272), dim3(256), 0, st, (const volatile unsigned long long*)nullptr, seq, scratch, t_prep);
        hipLaunchKernelGGL(k_score, dim3(256), dim3(512), 0, st, scratch, t_score);
        hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, st, scratch, d_rec, t_fold);
        wait(seq);
    };
    // capture the chain with the arguments read from the mapped word
    hipGraph_t graph;
    hipGraphExec_t exec;
    CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(k_prep, dim3(272), dim3(256), 0, st, (const volatile unsigned long long*)d_args, 0ull, scratch, t_prep);
    hipLaunchKernelGGL(k_score, dim3(256), dim3(512), 0, st, scratch, t_score);
    hipLaunchKernelGGL(k_fold, dim3(1), dim3(256), 0, st, scratch, d_rec, t_fold);
    CHECK(hipStreamEndCapture(st, &graph));
    CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipGraphExec_t exec_p;                                     // a second executable graph: its first node's arguments are set per step
    CHECK(hipGraphInstantiate(&exec_p, graph, nullptr, nullptr, 0));
    auto graph_step = [&]() {
        ++seq;
        h_args[0] = seq;
        CHECK(hipGraphLaunch(exec, st));
        wait(seq);
    };
    // the same with the first node's arguments set before every launch
    size_t nn = 0;
    CHECK(hipGraphGetNodes(graph, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    CHECK(hipGraphGetNodes(graph, nodes.data(), &nn));
    hipGraphNode_t prep_node = nullptr;
    for (auto n : nodes) {
        hipKernelNodeParams p{};
        if (hipGraphKernelNodeGetParams(n, &p) == hipSuccess && p.func == (void*)k_prep) prep_node = n;
    }
    auto graph_params_step = [&]() {
        ++seq;
        const volatile unsigned long long* a0 = nullptr;
        unsigned long long a1 = seq;
        unsigned* a2 = scratch;
        float a3 = t_prep;
        void* kargs[] = {(void*)&a0, (void*)&a1, (void*)&a2, (void*)&a3};
        hipKernelNodeParams p{};
        p.func = (void*)k_prep; p.gridDim = dim3(272); p.blockDim = dim3(256); p.sharedMemBytes = 0; p.kernelParams = kargs; p.extra = nullptr;
        CHECK(hipGraphExecKernelNodeSetParams(exec_p, prep_node, &p));
        CHECK(hipGraphLaunch(exec_p, st));
        wait(seq);
    };
    auto run = [&](const char* name, auto step) {
        for (int i = 0; i < 200; ++i) step();
        std::vector<double> us;
        for (int rep = 0; rep < 5; ++rep) {
            const auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 2000; ++i) step();
            us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000.0);
        }
        std::sort(us.begin(), us.end());
        fflush(stdout);
        printf("%-14s %.2f us per step (min of 5: %.2f, max %.2f); kernel bodies %.0f us\n", name, us[2], us[0], us[4], t_prep + t_score + t_fold);
    };
    setvbuf(stdout, nullptr, _IOLBF, 0);
    for (int round = 0; round < 2; ++round) {
        run("stream", stream_step);
        run("graph", graph_step);
        if (prep_node) run("graph+params", graph_params_step);
    }
    return 0;
}
