// rt_bvh.h — host-side builder of the primitive (sphere / rectangle) BVH that k_intersect stages into LDS.
//
// The reference accelerates closest-hit with a random-axis median-split BVH of trait objects
// (hitable.rs:158-241).  The GPU path keeps the RESULT of HitableList::hit (hitable.rs:117-132:
// the closest root, ties to the later object) but searches with its own structure: a SAH BVH2 over the
// world entries (exact sweep over three axes for ranges up to 4096, 16 bins above), single-entry
// leaves, boxes padded so that culling is conservative with respect to the exact Sphere::hit /
// XYRect::hit arithmetic, collapsed greedily into 4-wide nodes (collapse_bvh4) that one set of
// 7 LDS reads decides.  Culling never changes which entry wins, so results are identical to the
// brute-force list walk (tests/test_gpu_parity.py::test_bvh_equals_brute_force_on_adversarial_rays).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <vector>

namespace rt {

// tight axis-aligned bounds of one primitive (before padding)
struct PrimBox {
    float mn[3], mx[3];
};

struct HostBvh {
    std::vector<float4> a, b, c;
    std::vector<int4> d;
    uint32_t depth = 0;
};

// 4-wide collapse of a HostBvh: node = min_x[4], min_y[4], min_z[4], max_x[4], max_y[4], max_z[4]
// (one float4 per plane over the 4 children) + child ids (>= 0 inner, < 0 sphere ~id, INT_MIN empty).
struct HostBvh4 {
    std::vector<float4> p[6];
    std::vector<int4> id;
    uint32_t depth = 0;
};

namespace bvh_detail {

struct Box {
    float mn[3], mx[3];
    void reset() {
        for (int k = 0; k < 3; ++k) mn[k] = FLT_MAX, mx[k] = -FLT_MAX;
    }
    void grow(const Box& o) {
        for (int k = 0; k < 3; ++k) mn[k] = std::min(mn[k], o.mn[k]), mx[k] = std::max(mx[k], o.mx[k]);
    }
    float half_area() const {
        float e0 = mx[0] - mn[0], e1 = mx[1] - mn[1], e2 = mx[2] - mn[2];
        return e0 * e1 + e1 * e2 + e2 * e0;
    }
};

struct Builder {
    std::vector<Box> boxes;     // padded primitive boxes
    std::vector<float> cen;     // 3 per primitive: box centres (the SAH binning key)
    std::vector<uint32_t> order;
    HostBvh& out;
    bool median_only;
    size_t sweep_max = 4096; // ranges up to this size use the exact sweep SAH

    // `prim` = tight primitive bounds (sphere: c -+ r; rectangle: its plane slab)
    Builder(const std::vector<PrimBox>& prim, HostBvh& o, bool median) : out(o), median_only(median) {
        boxes.resize(prim.size());
        cen.resize(prim.size() * 3);
        order.resize(prim.size());
        for (size_t i = 0; i < prim.size(); ++i) {
            order[i] = (uint32_t)i;
            for (int k = 0; k < 3; ++k) {
                // pad: 2^-18 of the coordinate magnitude, >= 10x the fp32 error of the exact tests
                const float mag = std::max(std::fabs(prim[i].mn[k]), std::fabs(prim[i].mx[k]));
                const float pad = mag * 3.8146973e-06f + 1e-30f;
                boxes[i].mn[k] = prim[i].mn[k] - pad;
                boxes[i].mx[k] = prim[i].mx[k] + pad;
                if (!(boxes[i].mn[k] <= boxes[i].mx[k])) { // NaN/inf primitive: never cull it
                    boxes[i].mn[k] = -FLT_MAX, boxes[i].mx[k] = FLT_MAX;
                }
                cen[3 * i + (size_t)k] = 0.5f * prim[i].mn[k] + 0.5f * prim[i].mx[k];
            }
        }
    }

    Box range_box(size_t first, size_t count) const {
        Box b;
        b.reset();
        for (size_t i = first; i < first + count; ++i) b.grow(boxes[order[i]]);
        return b;
    }

    // returns child reference: >= 0 inner node index, < 0 ~sphere
    int build(size_t first, size_t count, uint32_t depth) {
        out.depth = std::max(out.depth, depth);
        if (count == 1) return ~(int)order[first];
        // centroid bounds
        float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (size_t i = first; i < first + count; ++i) {
            const float* c = &cen[3 * (size_t)order[i]];
            for (int k = 0; k < 3; ++k) cmn[k] = std::min(cmn[k], c[k]), cmx[k] = std::max(cmx[k], c[k]);
        }
        int axis = 0;
        float ext = cmx[0] - cmn[0];
        for (int k = 1; k < 3; ++k)
            if (cmx[k] - cmn[k] > ext) ext = cmx[k] - cmn[k], axis = k;
        auto cen = [&](uint32_t id) { return this->cen[3 * (size_t)id + (size_t)axis]; };
        size_t mid = first + count / 2;
        bool done = false;
        // Small ranges: exact sweep SAH over all three axes (every split position between centroid-sorted
        // primitives) instead of 16 bins on the widest axis; the binned search takes over above RT_SWEEP_MAX.
        if (!median_only && count > 2 && count <= sweep_max) {
            float best = FLT_MAX;
            int best_axis = -1;
            size_t best_mid = 0;
            std::vector<uint32_t> tmp(order.begin() + (long)first, order.begin() + (long)(first + count));
            std::vector<float> right(count);
            for (int ax = 0; ax < 3; ++ax) {
                std::stable_sort(tmp.begin(), tmp.end(), [&](uint32_t x, uint32_t y) { return this->cen[3 * (size_t)x + (size_t)ax] < this->cen[3 * (size_t)y + (size_t)ax]; });
                Box acc;
                acc.reset();
                for (size_t i = count; i-- > 1;) {
                    acc.grow(boxes[tmp[i]]);
                    right[i] = acc.half_area();
                }
                acc.reset();
                for (size_t i = 1; i < count; ++i) {
                    acc.grow(boxes[tmp[i - 1]]);
                    const float cost = acc.half_area() * (float)i + right[i] * (float)(count - i);
                    if (cost < best) best = cost, best_axis = ax, best_mid = i;
                }
            }
            if (best_axis >= 0) {
                std::stable_sort(order.begin() + (long)first, order.begin() + (long)(first + count), [&](uint32_t x, uint32_t y) {
                    return this->cen[3 * (size_t)x + (size_t)best_axis] < this->cen[3 * (size_t)y + (size_t)best_axis];
                });
                mid = first + best_mid;
                done = true;
            }
        }
        if (!done && !median_only && ext > 0.0f && count > 2) {
            const int NB = 16;
            Box bb[NB];
            size_t bn[NB] = {0};
            for (int k = 0; k < NB; ++k) bb[k].reset();
            const float scale = (float)NB / ext;
            auto bin_of = [&](uint32_t id) { return std::min(NB - 1, std::max(0, (int)((cen(id) - cmn[axis]) * scale))); };
            for (size_t i = first; i < first + count; ++i) {
                int k = bin_of(order[i]);
                bb[k].grow(boxes[order[i]]);
                ++bn[k];
            }
            float right_area[NB];
            size_t right_n[NB];
            Box acc;
            acc.reset();
            size_t n = 0;
            for (int k = NB - 1; k >= 1; --k) {
                if (bn[k]) acc.grow(bb[k]);
                n += bn[k];
                right_area[k] = n ? acc.half_area() : 0.0f;
                right_n[k] = n;
            }
            acc.reset();
            n = 0;
            float best = FLT_MAX;
            int best_k = -1;
            for (int k = 0; k < NB - 1; ++k) {
                if (bn[k]) acc.grow(bb[k]);
                n += bn[k];
                if (n == 0 || right_n[k + 1] == 0) continue;
                float cost = acc.half_area() * (float)n + right_area[k + 1] * (float)right_n[k + 1];
                if (cost < best) best = cost, best_k = k;
            }
            if (best_k >= 0) {
                auto it = std::stable_partition(order.begin() + (long)first, order.begin() + (long)(first + count),
                                                [&](uint32_t id) { return bin_of(id) <= best_k; });
                mid = (size_t)(it - order.begin());
                done = mid > first && mid < first + count;
            }
        }
        if (!done) {
            mid = first + count / 2;
            std::stable_sort(order.begin() + (long)first, order.begin() + (long)(first + count),
                             [&](uint32_t x, uint32_t y) { return cen(x) < cen(y); });
        }
        const int me = (int)out.a.size();
        out.a.push_back(float4{}), out.b.push_back(float4{}), out.c.push_back(float4{}), out.d.push_back(int4{});
        const Box lb = range_box(first, mid - first), rb = range_box(mid, first + count - mid);
        const int l = build(first, mid - first, depth + 1);
        const int r = build(mid, first + count - mid, depth + 1);
        out.a[(size_t)me] = make_float4(lb.mn[0], lb.mn[1], lb.mn[2], lb.mx[0]);
        out.b[(size_t)me] = make_float4(lb.mx[1], lb.mx[2], rb.mn[0], rb.mn[1]);
        out.c[(size_t)me] = make_float4(rb.mn[2], rb.mx[0], rb.mx[1], rb.mx[2]);
        out.d[(size_t)me] = make_int4(l, r, 0, 0);
        return me;
    }
};

} // namespace bvh_detail

// Builds the BVH over primitive bounds.  The root is always inner node 0 (a single primitive gets
// an empty right child).  `max_depth` bounds the traversal stack: if the SAH tree is deeper, a
// median-split tree (depth <= ceil(log2 n) + 1) is built instead.
inline void build_prim_bvh(const std::vector<PrimBox>& geo, uint32_t max_depth, HostBvh& out) {
    out = HostBvh();
    if (geo.empty()) return;
    if (geo.size() == 1) {
        bvh_detail::Builder b(geo, out, true);
        const bvh_detail::Box& bx = b.boxes[0];
        out.a.push_back(make_float4(bx.mn[0], bx.mn[1], bx.mn[2], bx.mx[0]));
        out.b.push_back(make_float4(bx.mx[1], bx.mx[2], FLT_MAX, FLT_MAX));
        out.c.push_back(make_float4(FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX));
        out.d.push_back(make_int4(~0, INT_MIN, 0, 0));
        out.depth = 1;
        return;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        out = HostBvh();
        bvh_detail::Builder b(geo, out, attempt == 1);
        b.build(0, geo.size(), 0);
        if (out.depth <= max_depth) return;
    }
}

namespace bvh_detail {
inline int collapse4(const HostBvh& b, int n, HostBvh4& o, uint32_t depth) {
    o.depth = std::max(o.depth, depth + 1);
    const int me = (int)o.id.size();
    for (auto& v : o.p) v.push_back(float4{});
    o.id.push_back(int4{});
    float mn[3][4], mx[3][4];
    int cid[4];
    int cnt = 0;
    auto add = [&](const float* bmn, const float* bmx, int child) {
        for (int k = 0; k < 3; ++k) mn[k][cnt] = bmn[k], mx[k][cnt] = bmx[k];
        cid[cnt++] = child;
    };
    auto boxes = [&](int node, float lmn[3], float lmx[3], float rmn[3], float rmx[3]) {
        const float4 A = b.a[(size_t)node], B = b.b[(size_t)node], C = b.c[(size_t)node];
        lmn[0] = A.x, lmn[1] = A.y, lmn[2] = A.z, lmx[0] = A.w, lmx[1] = B.x, lmx[2] = B.y;
        rmn[0] = B.z, rmn[1] = B.w, rmn[2] = C.x, rmx[0] = C.y, rmx[1] = C.z, rmx[2] = C.w;
    };
    // Greedy collapse: start from the two children of `n`; while there is room, replace the inner child with the
    // largest box by its own two children.  Compared with always taking the four grandchildren this fills the
    // nodes of unbalanced subtrees (a leaf next to a deep sibling) and opens the box a ray is most likely to enter.
    {
        float lmn[3], lmx[3], rmn[3], rmx[3];
        boxes(n, lmn, lmx, rmn, rmx);
        const int ch[2] = {b.d[(size_t)n].x, b.d[(size_t)n].y};
        if (ch[0] != INT_MIN) add(lmn, lmx, ch[0] >= 0 ? (1 << 30) + ch[0] : ch[0]);
        if (ch[1] != INT_MIN) add(rmn, rmx, ch[1] >= 0 ? (1 << 30) + ch[1] : ch[1]);
    }
    for (;;) {
        int pick = -1;
        float pick_area = -1.0f;
        for (int k = 0; k < cnt; ++k) {
            if (cid[k] < (1 << 30)) continue; // a leaf
            const float e0 = mx[0][k] - mn[0][k], e1 = mx[1][k] - mn[1][k], e2 = mx[2][k] - mn[2][k];
            const float area = e0 * e1 + e1 * e2 + e2 * e0;
            if (area > pick_area) pick_area = area, pick = k;
        }
        if (pick < 0) break;
        const int node = cid[pick] - (1 << 30);
        const int g[2] = {b.d[(size_t)node].x, b.d[(size_t)node].y};
        const int n_new = (g[0] != INT_MIN) + (g[1] != INT_MIN);
        if (cnt - 1 + n_new > 4) break;
        float gl[3], glx[3], gr[3], grx[3];
        boxes(node, gl, glx, gr, grx);
        // remove `pick`, append its children
        for (int k = pick; k + 1 < cnt; ++k) {
            cid[k] = cid[k + 1];
            for (int ax = 0; ax < 3; ++ax) mn[ax][k] = mn[ax][k + 1], mx[ax][k] = mx[ax][k + 1];
        }
        --cnt;
        if (g[0] != INT_MIN) add(gl, glx, g[0] >= 0 ? (1 << 30) + g[0] : g[0]);
        if (g[1] != INT_MIN) add(gr, grx, g[1] >= 0 ? (1 << 30) + g[1] : g[1]);
    }
    // leaves first: sphere-only scenes test a leaf child inside the node step (rt_kernels.h), and a hit there
    // tightens the limit the inner children that follow are culled against
    for (int i = 1; i < cnt; ++i)
        for (int j = i; j > 0 && cid[j] < 0 && cid[j - 1] >= 0; --j) {
            std::swap(cid[j], cid[j - 1]);
            for (int k = 0; k < 3; ++k) std::swap(mn[k][j], mn[k][j - 1]), std::swap(mx[k][j], mx[k][j - 1]);
        }
    for (int k = 0; k < cnt; ++k)
        if (cid[k] >= (1 << 30)) cid[k] = collapse4(b, cid[k] - (1 << 30), o, depth + 1);
    for (int k = cnt; k < 4; ++k) {
        for (int ax = 0; ax < 3; ++ax) mn[ax][k] = FLT_MAX, mx[ax][k] = -FLT_MAX;
        cid[k] = INT_MIN;
    }
    for (int ax = 0; ax < 3; ++ax) {
        o.p[ax][(size_t)me] = make_float4(mn[ax][0], mn[ax][1], mn[ax][2], mn[ax][3]);
        o.p[3 + ax][(size_t)me] = make_float4(mx[ax][0], mx[ax][1], mx[ax][2], mx[ax][3]);
    }
    o.id[(size_t)me] = make_int4(cid[0], cid[1], cid[2], cid[3]);
    return me;
}
} // namespace bvh_detail

inline void collapse_bvh4(const HostBvh& b, HostBvh4& out) {
    out = HostBvh4();
    if (b.a.empty()) return;
    bvh_detail::collapse4(b, 0, out, 0);
}

} // namespace rt
