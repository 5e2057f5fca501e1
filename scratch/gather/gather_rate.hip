// Microbenchmark: how fast can a CU fetch random 64-B records (one per lane, 4 x dwordx4), as k_extend's node phase does?
// variants: 0 = per-lane 4 x 16 B, independent indices; 1 = same, dependent chain (next index from the data);
//           2 = quad-cooperative (4 lanes read the 4 x 16 B of one record; 4 rounds serve the 4 lanes' records), independent;
//           3 = per-lane 1 x 16 B only (independent); 4 = per-lane 2 x 32 B ... (dwordx4 x 2, 32-B records)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
template <int V>
__global__ void __launch_bounds__(256) k_gather(const uint4 *tab, uint32_t mask, int iters, int active_lanes, uint32_t *out, int lds_share) {
    __shared__ uint4 lds_tab[256];
    if (V == 5 || V == 6) { lds_tab[threadIdx.x] = tab[threadIdx.x]; __syncthreads(); }
    const uint32_t tid = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63;
    uint32_t idx = hash32(tid * 2654435761u + 1u) & mask;
    uint32_t acc = 0;
    if (lane >= active_lanes) { out[tid] = 0; return; }
    for (int i = 0; i < iters; i++) {
        if (V == 0 || V == 1) {
            const uint4 *r = tab + (size_t)idx * 4;
            const uint4 a = r[0], b = r[1], c = r[2], d = r[3];
            acc ^= a.x ^ b.y ^ c.z ^ d.w;
            idx = (V == 1 ? hash32(acc + tid) : hash32(idx + i * 0x9e3779b9u + tid)) & mask;
        } else if (V == 2) {
            uint32_t mine = 0;
            const uint32_t l0 = __builtin_amdgcn_mov_dpp(idx, 0x00, 0xf, 0xf, true), l1 = __builtin_amdgcn_mov_dpp(idx, 0x55, 0xf, 0xf, true),
                           l2 = __builtin_amdgcn_mov_dpp(idx, 0xaa, 0xf, 0xf, true), l3 = __builtin_amdgcn_mov_dpp(idx, 0xff, 0xf, 0xf, true);
            const uint4 a0 = tab[(size_t)l0 * 4 + (lane & 3)], a1 = tab[(size_t)l1 * 4 + (lane & 3)], a2 = tab[(size_t)l2 * 4 + (lane & 3)],
                        a3 = tab[(size_t)l3 * 4 + (lane & 3)];
            mine ^= a0.x ^ a0.y ^ a1.z ^ a1.w ^ a2.x ^ a2.z ^ a3.y ^ a3.w;
            acc ^= mine;
            idx = hash32(idx + i * 0x9e3779b9u + tid) & mask;
        } else if (V == 5 || V == 6) {
            const bool in_lds = (hash32(idx ^ 0x5bd1e995u) & 1023u) < (uint32_t)lds_share; // share of 1024
            const uint32_t li = idx & 63u; // 64 records of 64 B = 4 KB of LDS
            uint4 a, b, c, d;
            if (V == 5) {
                const uint4 *r = in_lds ? (const uint4 *)(lds_tab + li * 4) : tab + (size_t)idx * 4; // generic pointer: flat loads
                a = r[0]; b = r[1]; c = r[2]; d = r[3];
            } else if (in_lds) {
                a = lds_tab[li * 4]; b = lds_tab[li * 4 + 1]; c = lds_tab[li * 4 + 2]; d = lds_tab[li * 4 + 3];
            } else {
                const uint4 *r = tab + (size_t)idx * 4;
                a = r[0]; b = r[1]; c = r[2]; d = r[3];
            }
            acc ^= a.x ^ b.y ^ c.z ^ d.w;
            idx = hash32(idx + i * 0x9e3779b9u + tid) & mask;
        } else if (V == 3) {
            const uint4 a = tab[(size_t)idx * 4];
            acc ^= a.x ^ a.w;
            idx = hash32(idx + i * 0x9e3779b9u + tid) & mask;
        } else if (V == 4) {
            const uint4 *r = tab + (size_t)idx * 4;
            const uint4 a = r[0], b = r[1];
            acc ^= a.x ^ b.y;
            idx = hash32(idx + i * 0x9e3779b9u + tid) & mask;
        }
    }
    out[tid] = acc;
}
int main(int argc, char **argv) {
    const int wg_per_cu = argc > 1 ? atoi(argv[1]) : 7;
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int n_cu = p.multiProcessorCount;
    const int grid = n_cu * wg_per_cu;
    uint32_t *out; CHECK(hipMalloc(&out, (size_t)grid * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (uint32_t log_rec : {15u, 17u}) { // 2 MB, 8 MB
        const size_t n_rec = 1ull << log_rec;
        uint4 *tab; CHECK(hipMalloc(&tab, n_rec * 64));
        std::vector<uint32_t> h(n_rec * 16);
        for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) ^ (uint32_t)(i >> 7);
        CHECK(hipMemcpy(tab, h.data(), n_rec * 64, hipMemcpyHostToDevice));
        for (int v : {0, 5, 6}) for (int share : {0, 307, 614}) for (int lanes : {64, 40}) {
            if (v == 0 && share) continue;
            const int iters = 2000;
            float best = 1e30f;
            for (int rep = 0; rep < 3; rep++) {
                CHECK(hipEventRecord(e0));
                switch (v) {
                case 0: hipLaunchKernelGGL(k_gather<0>, dim3(grid), dim3(256), 0, 0, tab, (uint32_t)n_rec - 1, iters, lanes, out, share); break;
                case 1: hipLaunchKernelGGL(k_gather<1>, dim3(grid), dim3(256), 0, 0, tab, (uint32_t)n_rec - 1, iters, lanes, out, share); break;
                case 2: hipLaunchKernelGGL(k_gather<2>, dim3(grid), dim3(256), 0, 0, tab, (uint32_t)n_rec - 1, iters, lanes, out, share); break;
                case 3: hipLaunchKernelGGL(k_gather<3>, dim3(grid), dim3(256), 0, 0, tab, (uint32_t)n_rec - 1, iters, lanes, out, share); break;
                case 4: hipLaunchKernelGGL(k_gather<4>, dim3(grid), dim3(256), 0, 0, tab, (uint32_t)n_rec - 1, iters, lanes, out, share); break;
                case 5: hipLaunchKernelGGL(k_gather<5>, dim3(grid), dim3(256), 0, 0, tab, (uint32_t)n_rec - 1, iters, lanes, out, share); break;
                case 6: hipLaunchKernelGGL(k_gather<6>, dim3(grid), dim3(256), 0, 0, tab, (uint32_t)n_rec - 1, iters, lanes, out, share); break;
                }
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            const double recs = (double)grid * 4 * lanes * iters;
            printf("table %5zu KB  variant %d  lds share %d/1024  lanes %2d  wg/cu %d: %.3f ms  %.2f Grec/s  %.1f Mrec/s/CU  wave-step %.0f clk/CU @2.4GHz\n", n_rec * 64 >> 10, v, share, lanes, wg_per_cu,
                   best, recs / best * 1e-6, recs / best * 1e-3 / n_cu, best * 1e-3 * 2.4e9 / ((double)wg_per_cu * 4 * iters));
        }
        CHECK(hipFree(tab));
    }
    return 0;
}
