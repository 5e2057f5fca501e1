// mi_wide8.h — host side of the 8-wide node format (-DMI_WIDE8; the device side is mi_trav8.h): collapse of the binary
// tree into 8-wide nodes, seats assigned by octant, boxes quantised to 8 bits on each node's own grid, primitives
// permuted so that the leaf seats of a node name consecutive slots.  Replaces, like collapse_bvh4 / quantize_nodes4,
// the reference's flatten + flatten_clone (bvh.cu:239-252,395-415) — result-level: the closest hit does not depend on
// the tree.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

struct Wide8Tree {
    std::vector<uint32_t> words;            // 20 per node (80 B), breadth first, root = node 0
    std::vector<uint32_t> tri_src, sph_src; // new triangle slot i holds what was triangle slot tri_src[i] (spheres: index within the class)
    int levels = 0;                         // levels of 8-wide nodes
};

// nodes2: validated binary tree (64 B per interior node: two child boxes, two refs; ref < 0 = leaf ~slot).  margin[a]:
// what every plane is moved outward on axis a before it is rounded outward to the node's grid (covers the rounding of the
// slab arithmetic; the same 2^-18 of the grid's range the 16-bit codes of the 4-wide nodes get).
static void build_wide8(const float *nodes2, int32_t root2, uint32_t n_tri, uint32_t n_sph, Wide8Tree &out) {
    out.words.clear(); out.tri_src.clear(); out.sph_src.clear(); out.levels = 0;
    if (root2 < 0) { // no interior node: a single primitive (or none) keeps its slot
        if (n_tri + n_sph == 1) { if (n_tri) out.tri_src.push_back(0); else out.sph_src.push_back(0); }
        return;
    }
    struct Box { float lo[3], hi[3]; };
    auto child = [&](int32_t node, int c, Box &b, int32_t &r) {
        const float *nd = nodes2 + 16 * (size_t)node;
        memcpy(b.lo, nd + 6 * c, 12); memcpy(b.hi, nd + 6 * c + 3, 12); memcpy(&r, nd + 12 + c, 4);
    };
    // scene extent (root's box) -> margins
    double margin[3];
    {
        Box a, b; int32_t ra, rb;
        child(root2, 0, a, ra); child(root2, 1, b, rb);
        for (int k = 0; k < 3; k++) {
            const double ext = (double)std::max(a.hi[k], b.hi[k]) - (double)std::min(a.lo[k], b.lo[k]);
            margin[k] = (ext > 0 && std::isfinite(ext) ? ext : 1.0) * (3.814697265625e-06 / 1.48);
        }
    }
    struct Todo { int32_t n2; int level; };
    std::vector<Todo> queue{{root2, 1}};   // node index = position in the queue (breadth first)
    out.words.reserve((size_t)(n_tri + n_sph) * 6);
    for (size_t qi = 0; qi < queue.size(); qi++) {
        const Todo t = queue[qi];
        out.levels = std::max(out.levels, t.level);
        // children: three binary levels below the node (a leaf met on the way stays a seat)
        Box b[8]; int32_t ref[8]; int n = 2;
        child(t.n2, 0, b[0], ref[0]); child(t.n2, 1, b[1], ref[1]);
        for (int lvl = 1; lvl < 3; lvl++) {
            Box nb[8]; int32_t nr[8]; int m = 0;
            for (int k = 0; k < n; k++) {
                if (ref[k] < 0) { nb[m] = b[k]; nr[m++] = ref[k]; }
                else { child(ref[k], 0, nb[m], nr[m]); m++; child(ref[k], 1, nb[m], nr[m]); m++; }
            }
            n = m; memcpy(b, nb, sizeof nb); memcpy(ref, nr, sizeof nr);
        }
        Box nb = b[0];
        for (int k = 1; k < n; k++) for (int a = 0; a < 3; a++) { nb.lo[a] = std::min(nb.lo[a], b[k].lo[a]); nb.hi[a] = std::max(nb.hi[a], b[k].hi[a]); }
        // seats by octant: greedily the (child, seat) pair whose centre lies farthest along the seat's diagonal
        int seat_of[8]; bool cu[8] = {false}, su[8] = {false};
        float cost[8][8];
        for (int k = 0; k < n; k++)
            for (int s = 0; s < 8; s++) {
                float c = 0;
                for (int a = 0; a < 3; a++) { const float cen = 0.5f * (b[k].lo[a] + b[k].hi[a]) - 0.5f * (nb.lo[a] + nb.hi[a]); c += ((s >> a) & 1) ? cen : -cen; }
                cost[k][s] = c;
            }
        for (int it = 0; it < n; it++) {
            int bk = -1, bs = -1; float best = -INFINITY;
            for (int k = 0; k < n; k++) if (!cu[k]) for (int s = 0; s < 8; s++) if (!su[s] && (bk < 0 || cost[k][s] > best)) { best = cost[k][s]; bk = k; bs = s; }
            cu[bk] = su[bs] = true; seat_of[bk] = bs;
        }
        int kid[8]; for (int s = 0; s < 8; s++) kid[s] = -1;
        for (int k = 0; k < n; k++) kid[seat_of[k]] = k;
        // grid of the node: origin below the box by the margin, 2^e the smallest step that reaches the upper face in 255 steps
        uint32_t w[20] = {0};
        float origin[3]; int eb[3]; double step[3];
        for (int a = 0; a < 3; a++) {
            const double lo = (double)nb.lo[a] - margin[a], hi = (double)nb.hi[a] + margin[a];
            float of = (float)lo; if ((double)of > lo) of = std::nextafterf(of, -INFINITY);
            origin[a] = of;
            int e = (int)std::ceil(std::log2(std::max(hi - (double)of, 1e-300) / 255.0));
            while (std::ceil((hi - (double)of) / std::ldexp(1.0, e)) > 255.0) e++;
            e = std::max(-100, std::min(100, e)); // coordinates are validated below 1e15 < 2^50; a flat axis ends at -100
            eb[a] = e & 0xff; step[a] = std::ldexp(1.0, e);
            memcpy(&w[a], &of, 4);
        }
        uint32_t imask = 0, tmask = 0, smask = 0;
        uint8_t q[6][8];
        for (int s = 0; s < 8; s++) {
            const int k = kid[s];
            for (int a = 0; a < 3; a++) {
                if (k < 0) { q[a][s] = 255; q[3 + a][s] = 0; continue; } // inverted: never hit
                const double ql = std::floor(((double)b[k].lo[a] - margin[a] - (double)origin[a]) / step[a]);
                const double qh = std::ceil(((double)b[k].hi[a] + margin[a] - (double)origin[a]) / step[a]);
                q[a][s] = (uint8_t)std::max(0.0, std::min(255.0, ql));
                q[3 + a][s] = (uint8_t)std::max(0.0, std::min(255.0, qh));
            }
            if (k < 0) continue;
            if (ref[k] >= 0) imask |= 1u << s;
            else if ((uint32_t)~ref[k] < n_tri) tmask |= 1u << s;
            else smask |= 1u << s;
        }
        w[3] = (uint32_t)eb[0] | (uint32_t)eb[1] << 8 | (uint32_t)eb[2] << 16 | imask << 24;
        w[4] = (uint32_t)queue.size();          // child_base: the interior children follow one another in seat order
        w[5] = (uint32_t)out.tri_src.size();    // tri_base
        w[6] = (uint32_t)out.sph_src.size();    // sph_base
        w[7] = tmask | smask << 8;
        for (int s = 0; s < 8; s++) {
            const int k = kid[s];
            if (k < 0) continue;
            if (ref[k] >= 0) queue.push_back({ref[k], t.level + 1});
            else if ((uint32_t)~ref[k] < n_tri) out.tri_src.push_back((uint32_t)~ref[k]);
            else out.sph_src.push_back((uint32_t)~ref[k] - n_tri);
        }
        for (int pl = 0; pl < 6; pl++) // lo.x lo.y lo.z hi.x hi.y hi.z, two words each
            for (int h = 0; h < 2; h++)
                w[8 + 2 * pl + h] = (uint32_t)q[pl][4 * h] | (uint32_t)q[pl][4 * h + 1] << 8 | (uint32_t)q[pl][4 * h + 2] << 16 | (uint32_t)q[pl][4 * h + 3] << 24;
        out.words.insert(out.words.end(), w, w + 20);
    }
}

// Primitive arrays gathered into the order the 8-wide tree wants (one thread per new slot).
struct Wide8Prims {
    const float4 *tri, *tri_normal, *sph;
    const uint32_t *tri_mat, *sph_mat;
    const uint8_t *tri_has_vmat;
    const int32_t *prim_id;
};
struct Wide8PrimsOut {
    float4 *tri, *tri_normal, *sph;
    uint32_t *tri_mat, *sph_mat;
    uint8_t *tri_has_vmat;
    int32_t *prim_id;
};
__global__ void __launch_bounds__(256)
k_wide8_permute(uint32_t n_tri, uint32_t n_sph, const uint32_t *tri_src, const uint32_t *sph_src, Wide8Prims in, Wide8PrimsOut out) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < n_tri) {
        const uint32_t o = tri_src[i];
        for (int k = 0; k < 3; k++) out.tri[3 * (size_t)i + k] = in.tri[3 * (size_t)o + k];
        out.tri_normal[i] = in.tri_normal[o];
        for (int k = 0; k < 3; k++) out.tri_mat[3 * (size_t)i + k] = in.tri_mat[3 * (size_t)o + k];
        out.tri_has_vmat[i] = in.tri_has_vmat[o];
        out.prim_id[i] = in.prim_id[o];
    } else if (i < n_tri + n_sph) {
        const uint32_t r = i - n_tri, o = sph_src[r];
        out.sph[r] = in.sph[o];
        out.sph_mat[r] = in.sph_mat[o];
        out.prim_id[i] = in.prim_id[n_tri + o];
    }
}
