// Algorithmic experiment (CPU, counts only): node visits and primitive tests per ray of W-wide trees collapsed from the
// same binary tree, for primary rays and for cosine-distributed bounce rays leaving the primary hit points.
// Built as a shared object and driven from scratch/bvh8/sim.py.  Not product code, not the oracle.
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <random>
struct Box { float lo[3], hi[3]; };
struct WNode { int n; Box b[8]; int32_t ref[8]; };
static float area(const Box &b) { float x = b.hi[0]-b.lo[0], y = b.hi[1]-b.lo[1], z = b.hi[2]-b.lo[2]; return x*y+y*z+z*x; }
struct Sim {
    const float *n2; int32_t root2; const float *tri; int n_tri;
    std::vector<WNode> nodes; int32_t root;
    int build(int32_t r2, int W, int mode) { // returns index
        int idx = (int)nodes.size(); nodes.emplace_back();
        Box b[8]; int32_t ref[8]; int n = 0;
        auto child = [&](int32_t node, int c, Box &o, int32_t &r) { const float *nd = n2 + 16*(size_t)node; memcpy(o.lo, nd+6*c, 12); memcpy(o.hi, nd+6*c+3, 12); memcpy(&r, nd+12+c, 4); };
        child(r2, 0, b[0], ref[0]); child(r2, 1, b[1], ref[1]); n = 2;
        if (mode == 0) { // rigid levels: open every interior seat, level by level, while it fits
            for (int lvl = 1; (1 << lvl) < W; lvl++) {
                Box nb[8]; int32_t nr[8]; int m = 0;
                for (int k = 0; k < n; k++) { if (ref[k] < 0) { nb[m] = b[k]; nr[m++] = ref[k]; } else { child(ref[k], 0, nb[m], nr[m]); m++; child(ref[k], 1, nb[m], nr[m]); m++; } }
                n = m; memcpy(b, nb, sizeof nb); memcpy(ref, nr, sizeof nr);
            }
        } else { // greedy: open the interior seat of the largest area until W seats
            while (n < W) { int open = -1; float best = -1; for (int k = 0; k < n; k++) if (ref[k] >= 0 && area(b[k]) > best) { best = area(b[k]); open = k; }
                if (open < 0) break; int32_t r = ref[open]; child(r, 0, b[open], ref[open]); child(r, 1, b[n], ref[n]); n++; }
        }
        for (int k = 0; k < n; k++) if (ref[k] >= 0) { int ci = build(ref[k], W, mode); ref[k] = ci; }
        WNode &nd = nodes[idx]; nd.n = n; memcpy(nd.b, b, sizeof b); memcpy(nd.ref, ref, sizeof ref);
        return idx;
    }
    bool tri_test(int s, const float *o, const float *d, float &t, float *nrm) {
        const float *r = tri + 16*(size_t)s; const float *v0 = r, *ui = r+4, *vi = r+8, *ni = r+12;
        float p[3] = {o[0]-v0[0], o[1]-v0[1], o[2]-v0[2]};
        float pz = ni[0]*p[0]+ni[1]*p[1]+ni[2]*p[2], dz = ni[0]*d[0]+ni[1]*d[1]+ni[2]*d[2];
        if ((pz < 0) == (dz < 0)) return false;
        float tt = -pz/dz; if (tt > t) return false;
        float h[3] = {p[0]+tt*d[0], p[1]+tt*d[1], p[2]+tt*d[2]};
        float u = ui[0]*h[0]+ui[1]*h[1]+ui[2]*h[2], v = vi[0]*h[0]+vi[1]*h[1]+vi[2]*h[2];
        if (u < 0 || u > 1 || v < 0 || v > 1 || u+v > 1) return false;
        t = tt; float l = std::sqrt(ni[0]*ni[0]+ni[1]*ni[1]+ni[2]*ni[2]); nrm[0] = ni[0]/l; nrm[1] = ni[1]/l; nrm[2] = ni[2]/l; return true;
    }
    // closest hit; counts node visits and primitive tests
    bool trace(const float *o, const float *d, float &t, float *nrm, uint64_t &nv, uint64_t &pt) {
        float inv[3] = {1/d[0], 1/d[1], 1/d[2]};
        int32_t stack[256]; float sk[256]; int sp = 0; bool hit = false; t = 1e30f;
        int32_t cur = root; float curk = 0;
        if (root2 < 0) return false;
        for (;;) {
            if (cur >= 0) {
                if (curk <= t) {
                nv++;
                const WNode &nd = nodes[cur];
                float k[8]; int32_t r[8]; int m = 0;
                for (int c = 0; c < nd.n; c++) {
                    float t0 = -1e30f, t1 = t;
                    for (int a = 0; a < 3; a++) { float x = (nd.b[c].lo[a]-o[a])*inv[a], y = (nd.b[c].hi[a]-o[a])*inv[a]; t0 = std::max(t0, std::min(x,y)); t1 = std::min(t1, std::max(x,y)); }
                    if (t0 <= t1 && t1 > 0) { k[m] = t0; r[m++] = nd.ref[c]; }
                }
                // sort descending by entry distance, push far first
                for (int i = 1; i < m; i++) for (int j = i; j > 0 && k[j-1] < k[j]; j--) { std::swap(k[j-1], k[j]); std::swap(r[j-1], r[j]); }
                for (int i = 0; i < m; i++) { stack[sp] = r[i]; sk[sp++] = k[i]; }
                }
            } else {
                pt++;
                if (tri_test(~cur, o, d, t, nrm)) hit = true;
            }
            if (sp == 0) break;
            sp--; cur = stack[sp]; curk = sk[sp];
        }
        return hit;
    }
};
extern "C" void sim_run(const float *nodes2, int32_t root2, const float *tri, int n_tri, const float *rays, int64_t n_rays, int W, int mode, int cull_pop,
                        double *out /* nv0, pt0, n0, nv1, pt1, n1, nodes */) {
    Sim s{nodes2, root2, tri, n_tri}; s.nodes.reserve(n_tri); s.root = s.build(root2, W, mode);
    std::mt19937 rng(1); std::uniform_real_distribution<float> U(0, 1);
    uint64_t nv0 = 0, pt0 = 0, nv1 = 0, pt1 = 0, n1 = 0;
    for (int64_t i = 0; i < n_rays; i++) {
        const float *o = rays + 6*i, *d = o + 3; float t, n[3];
        if (!s.trace(o, d, t, n, nv0, pt0)) continue;
        if (n[0]*d[0]+n[1]*d[1]+n[2]*d[2] > 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
        float h[3] = {o[0]+t*d[0]+1e-4f*n[0], o[1]+t*d[1]+1e-4f*n[1], o[2]+t*d[2]+1e-4f*n[2]};
        // cosine-ish hemisphere direction about n
        float u1 = U(rng), u2 = U(rng), ph = 6.2831853f*u1, ct = std::sqrt(u2), st = std::sqrt(1-u2);
        float a[3] = {std::fabs(n[0]) < 0.5f ? 1.f : 0.f, std::fabs(n[0]) < 0.5f ? 0.f : 1.f, 0};
        float bx[3] = {n[1]*a[2]-n[2]*a[1], n[2]*a[0]-n[0]*a[2], n[0]*a[1]-n[1]*a[0]}; float l = std::sqrt(bx[0]*bx[0]+bx[1]*bx[1]+bx[2]*bx[2]); for (float &q : bx) q /= l;
        float by[3] = {n[1]*bx[2]-n[2]*bx[1], n[2]*bx[0]-n[0]*bx[2], n[0]*bx[1]-n[1]*bx[0]};
        float d2[3]; for (int k = 0; k < 3; k++) d2[k] = st*std::cos(ph)*bx[k] + st*std::sin(ph)*by[k] + ct*n[k];
        float t2, n2v[3]; s.trace(h, d2, t2, n2v, nv1, pt1); n1++;
    }
    out[0] = (double)nv0; out[1] = (double)pt0; out[2] = (double)n_rays; out[3] = (double)nv1; out[4] = (double)pt1; out[5] = (double)n1; out[6] = (double)s.nodes.size();
}
