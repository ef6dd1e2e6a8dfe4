// dfusion_nanoflann.h -- the tie rule of the reference's k-NN.
//
// WarpField::KNN (/root/reference/kfusion/src/warp_field.cpp:247-251) asks the vendored nanoflann v1.2.3
// (kfusion/include/nanoflann/nanoflann.hpp) for the k nearest nodes.  The k smallest DISTANCES are unique, the node
// lists are not: when two nodes are exactly equidistant from the query, KNNResultSet::addPoint (:110-131) keeps the one it
// met first, and the order of meeting is the kd-tree's -- built by divideTree / middleSplit_ / planeSplit (:1042-1176)
// with leaf_max_size 10 (warp_field.cpp:20), walked near child first by searchLevel (:1200-1254).  On a node set
// sampled from a pixel grid such ties are common, and at rank k they decide WHICH node is blended.
//
// The k-NN kernels of dfusion_warp.hip rank candidates by distance only and raise a flag when an equal distance could
// matter (topk_insert).  Flagged queries are then answered by df_nf_search below: the same tree (df_nf_build replays
// nanoflann's build on the host, sequentially, as nanoflann does -- the partition order depends on every swap) and the
// same traversal, bound arithmetic and result-set rule, so the answer is nanoflann's own.  Unflagged queries have a
// unique answer, which is what the fast path returns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#define DF_NF_LEAF_MAX 10       // KDTreeSingleIndexAdaptorParams(10), warp_field.cpp:20
#define DF_NF_MAX_DEPTH 60      // search stack bound; deeper trees (pathological clustering) switch the tie rule off

// One tree node, 24 bytes.  feat < 0: leaf over vind[a, b).  Otherwise a / b are the low / high child and lo / hi the
// split values nanoflann calls divlow / divhigh (the children's tight bounds along `feat`).
struct DfNfNode { int a, b, feat; float lo, hi; int pad; };

struct DfNfView {
    const DfNfNode* nodes;      // null: tie rule unavailable (fall back to index order)
    const uint16_t* vind;       // nanoflann's permuted index array
    float blo[3], bhi[3];       // root bounding box
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
        nodes[me].a = c1; nodes[me].b = c2; nodes[me].feat = feat;
        nodes[me].lo = lhi[feat]; nodes[me].hi = rlo[feat];
        for (int c = 0; c < 3; ++c) { lo[c] = rlo[c] < llo[c] ? rlo[c] : llo[c]; hi[c] = lhi[c] < rhi[c] ? rhi[c] : lhi[c]; }
        return me;
    }
    // buildIndex :855-866.  Returns false for trees the device search stack cannot hold.
    bool build(const float* pos4, int M, float blo[3], float bhi[3])
    {
        pos = pos4; depth = 0;
        vind.resize(M); nodes.clear(); nodes.reserve(2 * (size_t)M / 5 + 8);
        for (int i = 0; i < M; ++i) vind[i] = i;
        for (int c = 0; c < 3; ++c) blo[c] = bhi[c] = pt(0, c);
        for (int k = 1; k < M; ++k)
            for (int c = 0; c < 3; ++c) { const float v = pt(k, c); if (v < blo[c]) blo[c] = v; if (v > bhi[c]) bhi[c] = v; }
        divide(0, M, blo, bhi, 1);
        return depth <= DF_NF_MAX_DEPTH;
    }
};

// ------------------------------------------------------------------------------------------------ device: search
template <int K> struct DfNfResult { float d[K]; int i[K]; };

// findNeighbors :903-917 + searchLevel :1200-1254 + KNNResultSet :92-137 for one query, recursion unrolled onto a stack:
//   tag 0 = descend into `node` with lower bound `md`;  tag 1 = back from a near child: decide about the far child;
//   tag 2 = back from a far child: restore dists[feat].
// Kept out of line: it runs for the few queries with an exact tie, and must not cost the callers registers.
template <int K>
__device__ __noinline__ DfNfResult<K> df_nf_search(const DfNfNode* __restrict__ nodes, const uint16_t* __restrict__ vind,
                                                   const float4* __restrict__ pos, float blo0, float blo1, float blo2, float bhi0,
                                                   float bhi1, float bhi2, float qx, float qy, float qz)
{
    DfNfResult<K> R;
    int count = 0;
#pragma unroll
    for (int i = 0; i < K; ++i) { R.d[i] = 0.f; R.i[i] = 0; }
    R.d[K - 1] = 3.402823466e+38f;                                   // init(): dists[capacity-1] = max()
    const float q[3] = {qx, qy, qz};
    const float blo[3] = {blo0, blo1, blo2}, bhi[3] = {bhi0, bhi1, bhi2};
    float dists[3] = {0.f, 0.f, 0.f};
    float md0 = 0.f;
    for (int c = 0; c < 3; ++c) {                                    // computeInitialDistances :1179-1196
        if (q[c] < blo[c]) { dists[c] = (q[c] - blo[c]) * (q[c] - blo[c]); md0 += dists[c]; }
        if (q[c] > bhi[c]) { dists[c] = (q[c] - bhi[c]) * (q[c] - bhi[c]); md0 += dists[c]; }
    }
    struct Frame { int node; float md; float cut; int tag_feat; };   // tag_feat = tag * 4 + feat
    Frame st[DF_NF_MAX_DEPTH + 4];                                   // at most depth + 1 frames are live
    int sp = 0;
    st[sp++] = Frame{0, md0, 0.f, 0};
    while (sp > 0) {
        const Frame f = st[--sp];
        const int tag = f.tag_feat >> 2, feat = f.tag_feat & 3;
        if (tag == 2) { dists[feat] = f.cut; continue; }             // (cut holds the saved value)
        if (tag == 1) {
            const float dst = dists[feat];
            const float md = f.md + f.cut - dst;
            if (md * 1.0f <= R.d[K - 1]) {                           // epsError = 1 + eps, eps = 0
                dists[feat] = f.cut;
                st[sp++] = Frame{0, 0.f, dst, 2 * 4 + feat};
                st[sp++] = Frame{f.node, md, 0.f, 0};
            }
            continue;
        }
        const DfNfNode n = nodes[f.node];
        if (n.feat < 0) {
            const float worst = R.d[K - 1];                          // worstDist() read once per leaf
            for (int i = n.a; i < n.b; ++i) {
                const int index = vind[i];
                const float4 p = pos[index];
                const float d0 = q[0] - p.x, d1 = q[1] - p.y, d2 = q[2] - p.z;
                const float dist = d0 * d0 + d1 * d1 + d2 * d2;      // knn_point_cloud.hpp:25-31
                if (dist < worst) {                                  // addPoint :110-131
                    int j = count;
                    for (; j > 0; --j) {
                        if (R.d[j - 1] > dist) { if (j < K) { R.d[j] = R.d[j - 1]; R.i[j] = R.i[j - 1]; } }
                        else break;
                    }
                    if (j < K) { R.d[j] = dist; R.i[j] = index; }
                    if (count < K) ++count;
                }
            }
            continue;
        }
        const float val = q[n.feat];
        const float diff1 = val - n.lo, diff2 = val - n.hi;
        int best, other; float cut;
        if ((diff1 + diff2) < 0) { best = n.a; other = n.b; cut = (val - n.hi) * (val - n.hi); }
        else { best = n.b; other = n.a; cut = (val - n.lo) * (val - n.lo); }
        st[sp++] = Frame{other, f.md, cut, 1 * 4 + n.feat};
        st[sp++] = Frame{best, f.md, 0.f, 0};
    }
    return R;
}
