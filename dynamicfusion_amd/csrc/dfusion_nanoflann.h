// dfusion_nanoflann.h -- the tie rule of the reference's k-NN.
//
// WarpField::KNN (/root/reference/kfusion/src/warp_field.cpp:247-251) asks the vendored nanoflann v1.2.3
// (kfusion/include/nanoflann/nanoflann.hpp) for the k nearest nodes.  The k smallest DISTANCES are unique, the node
// lists are not: when two nodes are exactly equidistant from the query, KNNResultSet::addPoint (:110-131) keeps the one it
// met first, and the order of meeting is the kd-tree's -- built by divideTree / middleSplit_ / planeSplit (:1042-1176)
// with leaf_max_size 10 (warp_field.cpp:20), walked near child first by searchLevel (:1200-1254).  On a node set
// sampled from a pixel grid such ties are common, and at rank k they decide WHICH node is blended.
//
// nanoflann's answer is therefore the first k nodes in the order (distance, visit order of THIS query's tree walk): a node
// it never visits (pruned subtree) is strictly farther than the k-th, and a node arriving with a distance already in the
// set goes behind it (strict `>` in addPoint, strict `<` against the leaf's worst distance).
//
// The k-NN kernels of dfusion_warp.hip rank candidates by distance; when (and only when) two distances are EQUAL they ask
// df_nf_visited_before() which of the two nodes the reference's walk meets first.  That needs no search: the walk visits
// the near child of every tree node first, so the order of two nodes is decided at the tree node where their root paths
// part (by the query's side of that node's split, exactly searchLevel's `(diff1 + diff2) < 0`), or, inside one leaf, by
// their position in nanoflann's permuted index array.  DfNfBuild replays nanoflann's build on the host -- sequentially, as
// nanoflann does: the partition order depends on every swap -- whenever the node set changes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#define DF_NF_LEAF_MAX 10       // KDTreeSingleIndexAdaptorParams(10), warp_field.cpp:20

// One tree node, 24 bytes.  feat < 0: leaf over vind[a, b).  Otherwise a / b are the low / high child, `mid` the first vind
// position of the high child (the low child covers the positions before it) and lo / hi the split values nanoflann calls
// divlow / divhigh (the children's tight bounds along `feat`).
struct DfNfNode { int a, b, feat; float lo, hi; int mid; };

struct DfNfView {
    const DfNfNode* nodes;      // null: tie rule unavailable (equal distances keep scan order)
    const uint16_t* vpos;       // vpos[node id] = position of the node in nanoflann's permuted index array vind
};

// ------------------------------------------------------------------------------------------------ host: build
struct DfNfBuild {
    const float* pos;           // M x 4 floats (xyz used), host
    std::vector<int> vind;
    std::vector<DfNfNode> nodes;
    int depth;
    float pt(int idx, int c) const { return pos[4 * (size_t)idx + c]; }

    void span(const int* ind, int count, int c, float& mn, float& mx) const      // computeMinMax :1094-1103
    {
        mn = mx = pt(ind[0], c);
        for (int i = 1; i < count; ++i) { const float v = pt(ind[i], c); if (v < mn) mn = v; if (v > mx) mx = v; }
    }
    // planeSplit :1151-1176, with nanoflann's unsigned index arithmetic (`right &&`, `!right`)
    void partition(int* ind, int count, int c, float cut, int& lim1, int& lim2) const
    {
        size_t l = 0, r = (size_t)count - 1;
        for (;;) {
            while (l <= r && pt(ind[l], c) < cut) ++l;
            while (r && l <= r && pt(ind[r], c) >= cut) --r;
            if (l > r || !r) break;
            std::swap(ind[l], ind[r]); ++l; --r;
        }
        lim1 = (int)l;
        r = (size_t)count - 1;
        for (;;) {
            while (l <= r && pt(ind[l], c) <= cut) ++l;
            while (r && l <= r && pt(ind[r], c) > cut) --r;
            if (l > r || !r) break;
            std::swap(ind[l], ind[r]); ++l; --r;
        }
        lim2 = (int)l;
    }
    // divideTree :1042-1091 (lo / hi = the bounding box handed down, tightened on the way up)
    int divide(int left, int right, float* lo, float* hi, int level)
    {
        const int me = (int)nodes.size();
        nodes.push_back(DfNfNode{0, 0, -1, 0.f, 0.f, 0});
        if (level > depth) depth = level;
        if (right - left <= DF_NF_LEAF_MAX) {
            for (int c = 0; c < 3; ++c) lo[c] = hi[c] = pt(vind[left], c);
            for (int k = left + 1; k < right; ++k)
                for (int c = 0; c < 3; ++c) {
                    const float v = pt(vind[k], c);
                    if (lo[c] > v) lo[c] = v;
                    if (hi[c] < v) hi[c] = v;
                }
            nodes[me].a = left; nodes[me].b = right;
            return me;
        }
        // middleSplit_ :1105-1140
        int* ind = vind.data() + left;
        const int count = right - left;
        const float EPS = 0.00001f;
        float max_span = hi[0] - lo[0];
        for (int c = 1; c < 3; ++c) { const float s = hi[c] - lo[c]; if (s > max_span) max_span = s; }
        float max_spread = -1.f; int feat = 0;
        for (int c = 0; c < 3; ++c) {
            const float s = hi[c] - lo[c];
            if (s > (1 - EPS) * max_span) {
                float mn, mx; span(ind, count, c, mn, mx);
                const float spread = mx - mn;
                if (spread > max_spread) { feat = c; max_spread = spread; }
            }
        }
        const float mid = (lo[feat] + hi[feat]) / 2;
        float mn, mx; span(ind, count, feat, mn, mx);
        const float cut = mid < mn ? mn : (mid > mx ? mx : mid);
        int lim1, lim2; partition(ind, count, feat, cut, lim1, lim2);
        const int idx = lim1 > count / 2 ? lim1 : (lim2 < count / 2 ? lim2 : count / 2);

        float llo[3] = {lo[0], lo[1], lo[2]}, lhi[3] = {hi[0], hi[1], hi[2]};
        lhi[feat] = cut;
        const int c1 = divide(left, left + idx, llo, lhi, level + 1);
        float rlo[3] = {lo[0], lo[1], lo[2]}, rhi[3] = {hi[0], hi[1], hi[2]};
        rlo[feat] = cut;
        const int c2 = divide(left + idx, right, rlo, rhi, level + 1);
        nodes[me].a = c1; nodes[me].b = c2; nodes[me].feat = feat; nodes[me].mid = left + idx;
        nodes[me].lo = lhi[feat]; nodes[me].hi = rlo[feat];
        for (int c = 0; c < 3; ++c) { lo[c] = rlo[c] < llo[c] ? rlo[c] : llo[c]; hi[c] = lhi[c] < rhi[c] ? rhi[c] : lhi[c]; }
        return me;
    }
    // buildIndex :855-866
    void build(const float* pos4, int M)
    {
        float blo[3], bhi[3];
        pos = pos4; depth = 0;
        vind.resize(M); nodes.clear(); nodes.reserve(2 * (size_t)M / 5 + 8);
        for (int i = 0; i < M; ++i) vind[i] = i;
        for (int c = 0; c < 3; ++c) blo[c] = bhi[c] = pt(0, c);
        for (int k = 1; k < M; ++k)
            for (int c = 0; c < 3; ++c) { const float v = pt(k, c); if (v < blo[c]) blo[c] = v; if (v > bhi[c]) bhi[c] = v; }
        divide(0, M, blo, bhi, 1);
    }
};

// ------------------------------------------------------------------------------------------------ device: visit order
// Does the reference's search for query (qx, qy, qz) meet node A before node B (A != B)?  searchLevel :1200-1254 descends into
// the near child first -- child1 iff (val - divlow) + (val - divhigh) < 0 -- and scans a leaf in vind order.
__device__ __forceinline__ bool df_nf_visited_before(const DfNfView& T, float qx, float qy, float qz, int A, int B)
{
    const int pa = T.vpos[A], pb = T.vpos[B];
    int node = 0;
    for (;;) {
        const DfNfNode n = T.nodes[node];
        if (n.feat < 0) return pa < pb;                              // same leaf: scan order
        const bool ha = pa >= n.mid, hb = pb >= n.mid;               // in the high child?
        if (ha != hb) {
            const float val = n.feat == 0 ? qx : (n.feat == 1 ? qy : qz);
            const bool low_first = ((val - n.lo) + (val - n.hi)) < 0;
            return ha != low_first;                                  // A is met first iff it sits in the child walked first
        }
        node = ha ? n.b : n.a;
    }
}
