// Probe: which way of putting a timing event into a captured graph works with the HIP runtime that is loaded?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void spin(float* p, int n) { float v = p[0]; for (int i = 0; i < n; ++i) v = v * 1.0001f + 1e-6f; p[0] = v; }
#define CK(x) do { hipError_t e = (x); printf("%-60s -> %s\n", #x, hipGetErrorString(e)); } while (0)
int main() {
    float* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipStream_t st; hipStreamCreate(&st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int ver = 0; hipRuntimeGetVersion(&ver); printf("runtime %d\n", ver);
    // (1) external flag during capture
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    spin<<<1, 1, 0, st>>>(d, 1000);
    CK(hipEventRecordWithFlags(a, st, hipEventRecordExternal));
    spin<<<1, 1, 0, st>>>(d, 2000000);
    CK(hipEventRecordWithFlags(b, st, hipEventRecordExternal));
    spin<<<1, 1, 0, st>>>(d, 1000);
    hipGraph_t g = nullptr; CK(hipStreamEndCapture(st, &g));
    if (g) {
        size_t n = 0; hipGraphGetNodes(g, nullptr, &n); printf("nodes %zu\n", n);
        hipGraphExec_t ge; CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int i = 0; i < 3; ++i) { CK(hipGraphLaunch(ge, st)); hipStreamSynchronize(st); float ms = -1; CK(hipEventElapsedTime(&ms, a, b)); printf("elapsed %f ms\n", ms); }
    }
    // (2) explicit graph surgery
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    spin<<<1, 1, 0, st>>>(d, 1000);
    spin<<<1, 1, 0, st>>>(d, 2000000);
    spin<<<1, 1, 0, st>>>(d, 1000);
    hipGraph_t g2 = nullptr; CK(hipStreamEndCapture(st, &g2));
    size_t n = 0; hipGraphGetNodes(g2, nullptr, &n);
    std::vector<hipGraphNode_t> nodes(n); hipGraphGetNodes(g2, nodes.data(), &n);
    printf("nodes %zu\n", n);
    // find the node with exactly one dependency and one dependent (the middle of the chain)
    for (size_t i = 0; i < n; ++i) {
        size_t nd = 0, nt = 0; hipGraphNodeGetDependencies(nodes[i], nullptr, &nd); hipGraphNodeGetDependentNodes(nodes[i], nullptr, &nt);
        if (nd == 1 && nt == 1) {
            hipGraphNode_t pred, succ; hipGraphNodeGetDependencies(nodes[i], &pred, &nd); hipGraphNodeGetDependentNodes(nodes[i], &succ, &nt);
            hipGraphNode_t ea, eb;
            CK(hipGraphAddEventRecordNode(&ea, g2, &pred, 1, a));
            CK(hipGraphAddDependencies(g2, &ea, &nodes[i], 1));
            CK(hipGraphAddEventRecordNode(&eb, g2, &nodes[i], 1, b));
            CK(hipGraphAddDependencies(g2, &eb, &succ, 1));
            break;
        }
    }
    hipGraphExec_t ge2; CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
    for (int i = 0; i < 3; ++i) { CK(hipGraphLaunch(ge2, st)); hipStreamSynchronize(st); float ms = -1; CK(hipEventElapsedTime(&ms, a, b)); printf("elapsed %f ms\n", ms); }
    return 0;
}
