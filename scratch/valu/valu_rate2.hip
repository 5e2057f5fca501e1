// Issue cost (shader cycles per wave64 instruction per SIMD, from s_memtime) of the VALU forms k_extend uses,
// at 4 waves per SIMD with 8 independent chains per wave.
// build: hipcc --offload-arch=gfx950 -O3 -o scratch/valu/valu_rate2 scratch/valu/valu_rate2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define BODY(ASM, ...)                                                           \
    for (int it = 0; it < iters; it++) {                                         \
        _Pragma("unroll") for (int r = 0; r < 8; r++) {                          \
            _Pragma("unroll") for (int i = 0; i < 8; i++) { asm volatile(ASM : __VA_ARGS__); } \
        }                                                                        \
    }

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, unsigned long long *cyc, int iters, float a, float b, uint32_t sa, uint32_t sb) {
    float x[8];
    uint32_t u[8];
    for (int i = 0; i < 8; i++) { x[i] = a + i + threadIdx.x; u[i] = (uint32_t)(threadIdx.x * 2654435761u + i); }
    float va = a + threadIdx.x * 1e-9f, vb = b + threadIdx.x * 1e-9f;
    uint32_t ua = sa + threadIdx.x, ub = sb ^ threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (OP == 0) BODY("v_fma_f32 %0, %0, %1, %2", "+v"(x[i]) : "v"(va), "v"(vb))
    if (OP == 1) BODY("v_fma_f32 %0, %0, %1, %2", "+v"(x[i]) : "s"(a), "v"(vb))
    if (OP == 2) BODY("v_fmac_f32 %0, %1, %2", "+v"(x[i]) : "v"(va), "v"(vb))
    if (OP == 3) BODY("v_mul_f32 %0, %0, %1", "+v"(x[i]) : "v"(va))
    if (OP == 4) BODY("v_perm_b32 %0, %0, %1, %2", "+v"(u[i]) : "v"(ua), "v"(ub))
    if (OP == 5) BODY("v_perm_b32 %0, %1, %0, %2", "+v"(u[i]) : "s"(sa), "v"(ub))
    if (OP == 6) BODY("v_perm_b32 %0, 0, %0, %1", "+v"(u[i]) : "s"(sb))
    if (OP == 7) BODY("v_max3_f32 %0, %0, %1, %2", "+v"(x[i]) : "v"(va), "v"(vb))
    if (OP == 8) BODY("v_max_f32 %0, %0, %1", "+v"(x[i]) : "v"(va))
    if (OP == 9) BODY("v_cmp_lt_f32 vcc, %0, %1", : "v"(x[i]), "v"(va) : "vcc")
    if (OP == 10) BODY("v_cndmask_b32 %0, %0, %1, vcc", "+v"(x[i]) : "v"(va) : "vcc")
    if (OP == 11) BODY("v_lshl_or_b32 %0, %0, 8, %1", "+v"(u[i]) : "v"(ua))
    if (OP == 12) BODY("v_lshl_or_b32 %0, %0, 8, %1", "+v"(u[i]) : "s"(sa))
    if (OP == 13) BODY("v_and_or_b32 %0, %0, %1, 64", "+v"(u[i]) : "s"(sa))
    if (OP == 14) BODY("v_lshlrev_b32 %0, 8, %0", "+v"(u[i]) :)
    if (OP == 15) BODY("v_and_b32 %0, %1, %0", "+v"(u[i]) : "s"(sa))
    if (OP == 16) BODY("v_bfe_u32 %0, %0, 8, 16", "+v"(u[i]) :)
    if (OP == 17) BODY("v_mov_b32 %0, %1", "=v"(u[i]) : "v"(ua))
    if (OP == 18) BODY("v_add_u32 %0, %0, %1", "+v"(u[i]) : "v"(ua))
    if (OP == 19) BODY("v_min3_f32 %0, %0, %1, %2", "+v"(x[i]) : "s"(a), "v"(vb))
    if (OP == 20) BODY("v_cndmask_b32 %0, %0, %1, %2", "+v"(x[i]) : "v"(va), "s"((unsigned long long)sa * 0x100000001ull))
    if (OP == 21) BODY("v_sub_f32 %0, %0, %1", "+v"(x[i]) : "v"(va))
    if (OP == 22) BODY("v_pk_fma_f32 %0, %0, %1, %2", "+v"(*(double *)&x[i & 6]) : "v"(*(double *)&va), "v"(*(double *)&vb))
    if (OP == 23) BODY("v_pk_mul_f32 %0, %0, %1", "+v"(*(double *)&x[i & 6]) : "v"(*(double *)&va))
    if (OP == 24) BODY("v_cvt_f32_ubyte1 %0, %0", "+v"(u[i]) :)
    if (OP == 25) BODY("v_alignbit_b32 %0, %0, %1, 8", "+v"(u[i]) : "v"(ua))
    if (OP == 26) BODY("v_mad_u32_u24 %0, %0, %1, %2", "+v"(u[i]) : "v"(ua), "v"(ub))
    if (OP == 27) BODY("v_med3_f32 %0, %0, %1, 1.0", "+v"(x[i]) : "s"(a))
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; i++) s += x[i] + (float)u[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char *name) {
    hipDeviceProp_t p;
    CHK(hipGetDeviceProperties(&p, 0));
    const int wg_per_cu = 4, grid = p.multiProcessorCount * wg_per_cu, iters = 4000;
    float *out;
    unsigned long long *cyc;
    CHK(hipMalloc(&out, (size_t)grid * 256 * 4));
    CHK(hipMalloc(&cyc, (size_t)grid * 4 * 8));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
        CHK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, out, cyc, iters, 1.0001f, 0.5f, 0x3F000000u, 0x07010c00u);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
    }
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)grid * 4);
    CHK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];
    const double inst_per_simd = (double)iters * 64 * wg_per_cu;
    printf("%-34s %.2f cycles per wave64 instruction per SIMD (s_memtime median, 4 waves/SIMD); wall %.3f ms -> clock %.2f GHz\n", name, med / inst_per_simd, ms,
           med / (ms * 1e-3) / 1e9);
    CHK(hipFree(out)); CHK(hipFree(cyc));
}

int main() {
    run<0>("v_fma_f32 v,v,v");
    run<1>("v_fma_f32 v,s,v");
    run<2>("v_fmac_f32 (VOP2)");
    run<3>("v_mul_f32");
    run<21>("v_sub_f32");
    run<4>("v_perm_b32 v,v,v");
    run<5>("v_perm_b32 s,v,v");
    run<6>("v_perm_b32 0,v,s");
    run<7>("v_max3_f32 v,v,v");
    run<19>("v_min3_f32 v,s,v");
    run<27>("v_med3_f32 v,s,1.0");
    run<8>("v_max_f32");
    run<9>("v_cmp_lt_f32 vcc");
    run<10>("v_cndmask_b32 vcc");
    run<20>("v_cndmask_b32 sgpr-pair (VOP3)");
    run<11>("v_lshl_or_b32 v,8,v");
    run<12>("v_lshl_or_b32 v,8,s");
    run<13>("v_and_or_b32 v,s,64");
    run<14>("v_lshlrev_b32");
    run<15>("v_and_b32 s,v");
    run<16>("v_bfe_u32 v,8,16");
    run<17>("v_mov_b32");
    run<18>("v_add_u32");
    run<22>("v_pk_fma_f32");
    run<23>("v_pk_mul_f32");
    run<24>("v_cvt_f32_ubyte1");
    run<25>("v_alignbit_b32 v,v,8");
    run<26>("v_mad_u32_u24 v,v,v");
    return 0;
}
