#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void k(unsigned *w, int K, int stride_words, int iters, unsigned *sink) {
    unsigned acc = 0;
    if (threadIdx.x == 0) {
        unsigned *p = w + (size_t)(blockIdx.x % K) * stride_words;
        for (int i = 0; i < iters; i++) acc += atomicAdd(p, 1u + (acc & 1u)); // dependent chain per block
    }
    if (acc == 0xffffffffu) *sink = acc;
}
int main() {
    unsigned *w, *sink;
    hipMalloc(&w, 64 << 20); hipMalloc(&sink, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    int Ks[] = {1, 2, 4, 8, 16, 64, 256};
    int strides[] = {16, 64, 1024, 65536};
    for (int s : strides) for (int K : Ks) {
        hipMemset(w, 0, 64 << 20);
        int blocks = 2048, iters = 64;
        k<<<blocks, 256>>>(w, K, s, 4, sink);
        hipEventRecord(a);
        k<<<blocks, 256>>>(w, K, s, iters, sink);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("stride %6d B  K=%3d : %.3f ms  -> %.1f atomics/us\n", s * 4, K, ms, blocks * (double)iters / (ms * 1e3));
    }
    return 0;
}
