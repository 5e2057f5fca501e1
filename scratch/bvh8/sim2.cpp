// Algorithmic experiment (CPU, counts only) for the 8-wide node format of round 3: nodes collapsed from the binary tree
// (greedy by area or rigid), child slots assigned by octant (Ylitie et al. 2017), child boxes quantised to 8 bits on the
// node's own grid (origin + 2^e * q, outward), traversal with one group entry per node (hit mask in priority order,
// leaves of a node tested before its children) — against the 4-wide, distance-sorted, exact-box walk the kernels do today.
// Not product code, not the oracle.
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <random>
struct Box { float lo[3], hi[3]; };
static float area(const Box &b) { float x = b.hi[0]-b.lo[0], y = b.hi[1]-b.lo[1], z = b.hi[2]-b.lo[2]; return x*y+y*z+z*x; }
struct WNode { int n; Box b[8]; int32_t ref[8]; };
struct Sim {
    const float *n2; int32_t root2; const float *tri; int n_tri;
    std::vector<WNode> nodes; int32_t root; int quant, order;
    // mode 2: SAH-optimal widening (Ylitie et al. 2017, 3.1): C[n][i] = least sum of wide-node areas when subtree n is given i+1 seats
    std::vector<float> C; std::vector<uint8_t> Ksplit; int Wdp = 0; std::vector<float> A2;
    float areaOf(int32_t n) { Box a, b; int32_t ra, rb; child(n, 0, a, ra); child(n, 1, b, rb); Box u; for (int k = 0; k < 3; k++) { u.lo[k] = std::min(a.lo[k], b.lo[k]); u.hi[k] = std::max(a.hi[k], b.hi[k]); } return area(u); }
    float cost(int32_t ref, int i) { return ref < 0 ? 0.0f : C[(size_t)ref * Wdp + std::min(i, Wdp - 1) - 0]; } // i = seats (1-based index i-1 below)
    void dp(int W, int n_nodes) {
        Wdp = W; C.assign((size_t)n_nodes * W, 0); Ksplit.assign((size_t)n_nodes * W, 0);
        // post-order over the binary tree
        std::vector<int32_t> order_; std::vector<int32_t> st{root2};
        while (!st.empty()) { int32_t n = st.back(); st.pop_back(); order_.push_back(n); Box b; int32_t r; for (int c = 0; c < 2; c++) { child(n, c, b, r); if (r >= 0) st.push_back(r); } }
        for (size_t q = order_.size(); q-- > 0;) {
            int32_t n = order_[q]; Box b; int32_t l, r; child(n, 0, b, l); child(n, 1, b, r);
            auto cc = [&](int32_t ref, int seats) { return ref < 0 ? 0.0f : C[(size_t)ref * W + (seats - 1)]; };
            // distribute(n, j): best split of j seats between the children (each at least 1)
            float dist[9]; uint8_t dk[9];
            for (int j = 2; j <= W; j++) { float best = 1e30f; int bk = 1; for (int k = 1; k < j; k++) { float v = cc(l, k) + cc(r, j - k); if (v < best) { best = v; bk = k; } } dist[j] = best; dk[j] = (uint8_t)bk; }
            float *Cn = &C[(size_t)n * W]; uint8_t *Kn = &Ksplit[(size_t)n * W];
            Cn[0] = dist[W] + areaOf(n); Kn[0] = 0;               // one seat: n becomes a wide node
            for (int i = 2; i <= W - 1; i++) { if (dist[i] < Cn[i - 2]) { Cn[i - 1] = dist[i]; Kn[i - 1] = dk[i]; } else { Cn[i - 1] = Cn[i - 2]; Kn[i - 1] = Kn[i - 2] ? Kn[i - 2] : 0; if (Cn[i-1] == Cn[0]) Kn[i-1] = 0; } }
            Cn[W - 1] = 0; // unused
            dkW[n] = dk[W];
        }
    }
    std::vector<uint8_t> dkW;
    // seats of subtree `ref` when it is given `seats` seats: appends (box, ref) pairs
    void expand(int32_t n2, int c, int seats, Box *b, int32_t *ref, int &n) {
        Box bb; int32_t r; child(n2, c, bb, r);
        if (r < 0 || seats == 1) { b[n] = bb; ref[n++] = r; return; }
        // how did C[r][seats-1] come about: walk down to the i that defined it
        int i = seats; const float *Cn = &C[(size_t)r * Wdp]; while (i > 1 && Cn[i - 1] == Cn[i - 2]) i--;
        if (i == 1) { b[n] = bb; ref[n++] = r; return; }
        int k = Ksplit[(size_t)r * Wdp + (i - 1)];
        expand(r, 0, k, b, ref, n); expand(r, 1, i - k, b, ref, n);
    }
    void child(int32_t node, int c, Box &o, int32_t &r) { const float *nd = n2 + 16*(size_t)node; memcpy(o.lo, nd+6*c, 12); memcpy(o.hi, nd+6*c+3, 12); memcpy(&r, nd+12+c, 4); }
    int build(int32_t r2, int W, int mode) {
        int idx = (int)nodes.size(); nodes.emplace_back();
        Box b[8]; int32_t ref[8]; int n = 0;
        child(r2, 0, b[0], ref[0]); child(r2, 1, b[1], ref[1]); n = 2;
        if (mode == 2) {
            n = 0; int k = dkW[r2];
            expand(r2, 0, k, b, ref, n); expand(r2, 1, W - k, b, ref, n);
        } else if (mode == 0) {
            for (int lvl = 1; (1 << lvl) < W; lvl++) {
                Box nb[8]; int32_t nr[8]; int m = 0;
                for (int k = 0; k < n; k++) { if (ref[k] < 0) { nb[m] = b[k]; nr[m++] = ref[k]; } else { child(ref[k], 0, nb[m], nr[m]); m++; child(ref[k], 1, nb[m], nr[m]); m++; } }
                n = m; memcpy(b, nb, sizeof nb); memcpy(ref, nr, sizeof nr);
            }
        } else {
            while (n < W) { int open = -1; float best = -1; for (int k = 0; k < n; k++) if (ref[k] >= 0 && area(b[k]) > best) { best = area(b[k]); open = k; }
                if (open < 0) break; int32_t r = ref[open]; child(r, 0, b[open], ref[open]); child(r, 1, b[n], ref[n]); n++; }
        }
        // node box, slot assignment by octant (greedy auction: best (child, slot) pair first)
        Box nb = b[0]; for (int k = 1; k < n; k++) for (int a = 0; a < 3; a++) { nb.lo[a] = std::min(nb.lo[a], b[k].lo[a]); nb.hi[a] = std::max(nb.hi[a], b[k].hi[a]); }
        Box sb[8]; int32_t sr[8]; for (int s = 0; s < 8; s++) sr[s] = 0x7fffffff;
        if (order == 1) {
            float cost[8][8]; bool cu[8] = {0}, su[8] = {0};
            for (int k = 0; k < n; k++) for (int s = 0; s < 8; s++) { float c = 0; for (int a = 0; a < 3; a++) { float cen = 0.5f*(b[k].lo[a]+b[k].hi[a]) - 0.5f*(nb.lo[a]+nb.hi[a]); c += ((s >> a) & 1) ? cen : -cen; } cost[k][s] = c; }
            for (int it = 0; it < n; it++) { int bk = -1, bs = -1; float best = -1e30f; for (int k = 0; k < n; k++) if (!cu[k]) for (int s = 0; s < 8; s++) if (!su[s] && cost[k][s] > best) { best = cost[k][s]; bk = k; bs = s; }
                cu[bk] = su[bs] = true; sb[bs] = b[bk]; sr[bs] = ref[bk]; }
        } else { for (int k = 0; k < n; k++) { sb[k] = b[k]; sr[k] = ref[k]; } }
        if (quant) { // 8-bit grid of the node: lo = floor, hi = ceil
            for (int a = 0; a < 3; a++) {
                float ext = nb.hi[a] - nb.lo[a]; int e = (int)std::ceil(std::log2(std::max(ext, 1e-30f) / 255.0f)); float sc = std::ldexp(1.0f, e);
                for (int s = 0; s < 8; s++) if (sr[s] != 0x7fffffff) {
                    float ql = std::floor((sb[s].lo[a] - nb.lo[a]) / sc), qh = std::ceil((sb[s].hi[a] - nb.lo[a]) / sc);
                    ql = std::max(0.0f, std::min(255.0f, ql)); qh = std::max(0.0f, std::min(255.0f, qh));
                    sb[s].lo[a] = nb.lo[a] + ql * sc; sb[s].hi[a] = nb.lo[a] + qh * sc;
                }
            }
        }
        for (int s = 0; s < 8; s++) if (sr[s] != 0x7fffffff && sr[s] >= 0) { int ci = build(sr[s], W, mode); sr[s] = ci; }
        WNode &nd = nodes[idx]; nd.n = 8; memcpy(nd.b, sb, sizeof sb); memcpy(nd.ref, sr, sizeof sr);
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
    // trav 0: distance-sorted, leaves through the stack (today's walk); trav 1: groups in octant order, a node's leaves first
    bool trace(const float *o, const float *d, float &t, float *nrm, uint64_t &nv, uint64_t &pt, uint64_t &steps, int trav) {
        float inv[3] = {1/d[0], 1/d[1], 1/d[2]};
        bool hit = false; t = 1e30f;
        if (root2 < 0) return false;
        auto boxhit = [&](const Box &b, float &t0) { t0 = -1e30f; float t1 = t; for (int a = 0; a < 3; a++) { float x = (b.lo[a]-o[a])*inv[a], y = (b.hi[a]-o[a])*inv[a]; t0 = std::max(t0, std::min(x,y)); t1 = std::min(t1, std::max(x,y)); } return t0 <= t1 && t1 > 0; };
        if (trav == 0) {
            int32_t stack[512]; int sp = 0; int32_t cur = root;
            for (;;) {
                steps++;
                if (cur >= 0) {
                    nv++;
                    const WNode &nd = nodes[cur];
                    float k[8]; int32_t r[8]; int m = 0;
                    for (int c = 0; c < 8; c++) { if (nd.ref[c] == 0x7fffffff) continue; float t0; if (boxhit(nd.b[c], t0)) { k[m] = t0; r[m++] = nd.ref[c]; } }
                    for (int i = 1; i < m; i++) for (int j = i; j > 0 && k[j-1] < k[j]; j--) { std::swap(k[j-1], k[j]); std::swap(r[j-1], r[j]); }
                    for (int i = 0; i < m; i++) stack[sp++] = r[i];
                } else { pt++; if (tri_test(~cur, o, d, t, nrm)) hit = true; }
                if (sp == 0) break;
                cur = stack[--sp];
            }
            return hit;
        }
        const int oct = (d[0] < 0 ? 1 : 0) | (d[1] < 0 ? 2 : 0) | (d[2] < 0 ? 4 : 0); // slot s lies towards +axis where bit set: nearest first = slots with s ^ oct ... smallest (s^oct) bits... use priority p = 7 - (s ^ oct)... see below
        struct Grp { int32_t node; unsigned mask; }; Grp stack[128]; int sp = 0;
        Grp cur{-1, 0}; int32_t next = root;
        for (;;) {
            if (next >= 0) { // visit node `next`
                steps++; nv++;
                const WNode &nd = nodes[next];
                unsigned hm = 0;
                for (int s = 0; s < 8; s++) { if (nd.ref[s] == 0x7fffffff) continue; float t0; if (boxhit(nd.b[s], t0)) hm |= 1u << s; }
                // leaves of this node first, in traversal order
                for (int p = 0; p < 8; p++) { int s = p ^ oct; if (!((hm >> s) & 1) || nd.ref[s] >= 0) continue; steps++; pt++; if (tri_test(~nd.ref[s], o, d, t, nrm)) hit = true; hm &= ~(1u << s); }
                if (cur.mask) stack[sp++] = cur;
                cur = Grp{next, hm};
                next = -1;
            }
            if (!cur.mask) { if (sp == 0) break; cur = stack[--sp]; }
            // nearest-first: slot s is on the + side of axis a when bit a set; a ray with d[a] < 0 comes from +: wants those first
            int pick = -1; for (int p = 0; p < 8; p++) { int s = p ^ oct; if ((cur.mask >> s) & 1) { pick = s; break; } }
            cur.mask &= ~(1u << pick);
            next = nodes[cur.node].ref[pick];
        }
        return hit;
    }
};
extern "C" void sim2_run(const float *nodes2, int32_t root2, const float *tri, int n_tri, const float *rays, int64_t n_rays, int W, int mode, int quant, int order, int trav,
                         double *out /* nv0, pt0, n0, nv1, pt1, n1, nodes, steps0, steps1 */) {
    Sim s{nodes2, root2, tri, n_tri}; s.quant = quant; s.order = order; s.nodes.reserve(n_tri);
    if (mode == 2) { s.dkW.assign(n_tri, 1); s.dp(W, n_tri - 1 + 1); }
    s.root = s.build(root2, W, mode);
    std::mt19937 rng(1); std::uniform_real_distribution<float> U(0, 1);
    uint64_t nv0 = 0, pt0 = 0, nv1 = 0, pt1 = 0, n1 = 0, st0 = 0, st1 = 0;
    for (int64_t i = 0; i < n_rays; i++) {
        const float *o = rays + 6*i, *d = o + 3; float t, n[3];
        if (!s.trace(o, d, t, n, nv0, pt0, st0, trav)) continue;
        if (n[0]*d[0]+n[1]*d[1]+n[2]*d[2] > 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
        float h[3] = {o[0]+t*d[0]+1e-4f*n[0], o[1]+t*d[1]+1e-4f*n[1], o[2]+t*d[2]+1e-4f*n[2]};
        float u1 = U(rng), u2 = U(rng), ph = 6.2831853f*u1, ct = std::sqrt(u2), st = std::sqrt(1-u2);
        float a[3] = {std::fabs(n[0]) < 0.5f ? 1.f : 0.f, std::fabs(n[0]) < 0.5f ? 0.f : 1.f, 0};
        float bx[3] = {n[1]*a[2]-n[2]*a[1], n[2]*a[0]-n[0]*a[2], n[0]*a[1]-n[1]*a[0]}; float l = std::sqrt(bx[0]*bx[0]+bx[1]*bx[1]+bx[2]*bx[2]); for (float &q : bx) q /= l;
        float by[3] = {n[1]*bx[2]-n[2]*bx[1], n[2]*bx[0]-n[0]*bx[2], n[0]*bx[1]-n[1]*bx[0]};
        float d2[3]; for (int k = 0; k < 3; k++) d2[k] = st*std::cos(ph)*bx[k] + st*std::sin(ph)*by[k] + ct*n[k];
        float t2, n2v[3]; s.trace(h, d2, t2, n2v, nv1, pt1, st1, trav); n1++;
    }
    out[0] = (double)nv0; out[1] = (double)pt0; out[2] = (double)n_rays; out[3] = (double)nv1; out[4] = (double)pt1; out[5] = (double)n1; out[6] = (double)s.nodes.size();
    out[7] = (double)st0; out[8] = (double)st1;
}
