// Issue rate of plain wave64 VALU instructions per SIMD on MI355X: how many cycles does one wave64
// instruction occupy a SIMD when 1, 2, 4, 8 waves per SIMD issue independent chains?
// build: hipcc --offload-arch=gfx950 -O3 -o scratch/valu/valu_rate scratch/valu/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a, float b) {
    float x[8];
    uint32_t u[8];
    for (int i = 0; i < 8; i++) { x[i] = a + i + threadIdx.x; u[i] = (uint32_t)(threadIdx.x * 2654435761u + i); }
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (OP == 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(0x07010c00u));
                if (OP == 2) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
                if (OP == 3) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[i]) : "v"(a), "v"(b) : "vcc");
                if (OP == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(*(double *)&x[i & 6]) : "v"(*(double *)&a), "v"(*(double *)&b));
                if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (OP == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[i]) : "v"(a));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i] + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
void run(const char *name, int per_inst) {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    float *out;
    CHK(hipMalloc(&out, (size_t)p.multiProcessorCount * 8 * 256 * 4 * 4));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int wg_per_cu : {1, 2, 4, 8}) { // 256 threads = 4 waves = 1 wave per SIMD per workgroup
        const int grid = p.multiProcessorCount * wg_per_cu;
        for (int rep = 0; rep < 2; rep++) {
            CHK(hipEventRecord(e0));
            hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
        }
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double inst_per_simd = (double)iters * 64 * per_inst * wg_per_cu; // wave-instructions issued on each SIMD
        const double cyc = ms * 1e-3 * (double)p.clockRate * 1e3;
        printf("%-22s waves/SIMD %d: %.3f ms, %.2f cycles per wave64 instruction per SIMD (at %d MHz nominal)\n", name, wg_per_cu, ms, cyc / inst_per_simd, p.clockRate / 1000);
    }
    CHK(hipFree(out));
}

int main() {
    run<0>("v_fma_f32", 1);
    run<1>("v_perm_b32", 1);
    run<2>("v_max3_f32", 1);
    run<3>("v_cmp+v_cndmask", 2);
    run<4>("v_pk_fma_f32", 1);
    run<5>("v_add_u32", 1);
    run<6>("v_mul_f32", 1);
    return 0;
}
