// Exact k-nearest-neighbour search on the unit sphere (chord distance) — the device
// counterpart of the reference's one-off index build
//   R: spatial.py:587-635  cKDTree(source_xyz).query(target_xyz, k)
// whose output feeds every gather kernel.  Seconds to minutes on one CPU thread for
// O1280 / O2560; here:
//   build : Morton keys of the source points -> radix sort (hipCUB) -> points gathered into
//           sorted order -> leaf buckets of 8 consecutive points -> implicit balanced
//           binary tree of axis-aligned boxes over the leaves (bottom-up, one launch per level)
//   query : one lane per target, depth-first traversal with an explicit stack, nearer child
//           first, subtree skipped when its box is farther than the current k-th best.
//
// Exactness / parity.  Coordinates come from the host (numpy deg2rad / cos / sin exactly as
// R: spatial.py:132-167) and squared distances are accumulated as scipy's
// sqeuclidean_distance_double does for 3-D points: s = 0; s += dx*dx; s += dy*dy; s += dz*dz
// in float64 without contraction — so the returned d^2 are BIT-IDENTICAL to cKDTree's and
// the neighbour lists agree wherever the (k+1) smallest distances are distinct.  Exact ties
// (cKDTree leaves their order to its traversal: which of two equidistant points it meets first
// depends on std::nth_element's permutation during ITS tree build) are ordered here by the lower
// source index; the host wrapper (interp.nearest_grid_points_device) asks for one neighbour more
// than needed, finds the rows with an exact tie among those k+1 distances and lets cKDTree itself
// decide these rows, so the index table it returns is identical to the reference's.
// The box lower bound is computed with the same monotone operations, so pruning never drops
// a point that could enter the list.
#include "atx_common.hpp"

#include <hipcub/hipcub.hpp>

namespace atx {

constexpr int kLeaf = 8;      // source points per leaf bucket
constexpr int kMaxK = 17;     // neighbours kept per target: 16 + the look-ahead neighbour of the tie detection
constexpr int kStack = 64;    // >= 2 * tree depth (depth <= 28 for 2^31 points)

struct KnnHeader {
    int64_t n_src;
    int64_t n_leaves;
    int64_t n_pow2;  // leaves padded to a power of two: node i has children 2i, 2i+1; leaf j is node n_pow2 + j
    int64_t off_xyz, off_order, off_boxes, off_keys_in, off_keys_out, off_vals_in, off_tmp;
    size_t tmp_bytes;
    size_t total_bytes;
};

static size_t align256(size_t x) { return (x + 255) & ~size_t(255); }

static KnnHeader knn_layout(int64_t n_src) {
    KnnHeader h{};
    h.n_src = n_src;
    h.n_leaves = (n_src + kLeaf - 1) / kLeaf;
    int64_t p = 1;
    while (p < h.n_leaves) p <<= 1;
    h.n_pow2 = p;
    size_t off = align256(sizeof(KnnHeader));
    h.off_xyz = off;        off = align256(off + (size_t)n_src * 3 * sizeof(double));
    h.off_order = off;      off = align256(off + (size_t)n_src * sizeof(int32_t));
    h.off_boxes = off;      off = align256(off + (size_t)2 * p * 6 * sizeof(double));
    h.off_keys_in = off;    off = align256(off + (size_t)n_src * sizeof(uint64_t));
    h.off_keys_out = off;   off = align256(off + (size_t)n_src * sizeof(uint64_t));
    h.off_vals_in = off;    off = align256(off + (size_t)n_src * sizeof(int32_t));
    h.off_tmp = off;
    size_t tmp = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, tmp, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr,
                                             (int32_t*)nullptr, (int)n_src, 0, 63);
    h.tmp_bytes = tmp;
    h.total_bytes = align256(off + tmp);
    return h;
}

__device__ __forceinline__ uint64_t spread21(uint64_t v) {
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__global__ void __launch_bounds__(kBlock)
knn_keys_kernel(const double* __restrict__ xyz, int64_t n, uint64_t* __restrict__ keys, int32_t* __restrict__ vals) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t q[3];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            double v = (xyz[i * 3 + d] + 1.0) * 0.5;  // [-1, 1] -> [0, 1]
            v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            if (!(v == v)) v = 0.0;
            q[d] = (uint64_t)(v * 2097151.0);
        }
        keys[i] = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
        vals[i] = (int32_t)i;
    }
}

__global__ void __launch_bounds__(kBlock)
knn_gather_kernel(const double* __restrict__ xyz, const int32_t* __restrict__ order, int64_t n, double* __restrict__ sorted) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const int64_t s = order[i];
        sorted[i * 3 + 0] = xyz[s * 3 + 0];
        sorted[i * 3 + 1] = xyz[s * 3 + 1];
        sorted[i * 3 + 2] = xyz[s * 3 + 2];
    }
}

__global__ void __launch_bounds__(kBlock)
knn_leaf_boxes_kernel(const double* __restrict__ sorted, int64_t n_src, int64_t n_pow2, double* __restrict__ boxes) {
    for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n_pow2; j += (int64_t)gridDim.x * kBlock) {
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        const int64_t p0 = j * kLeaf, p1 = min(n_src, p0 + kLeaf);
        for (int64_t p = p0; p < p1; ++p) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const double v = sorted[p * 3 + d];
                lo[d] = v < lo[d] ? v : lo[d];
                hi[d] = v > hi[d] ? v : hi[d];
            }
        }
        double* b = boxes + (n_pow2 + j) * 6;
#pragma unroll
        for (int d = 0; d < 3; ++d) { b[d] = lo[d]; b[3 + d] = hi[d]; }
    }
}

__global__ void __launch_bounds__(kBlock)
knn_merge_boxes_kernel(double* __restrict__ boxes, int64_t first, int64_t count) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < count; i += (int64_t)gridDim.x * kBlock) {
        const int64_t node = first + i;
        const double* a = boxes + (2 * node) * 6;
        const double* c = boxes + (2 * node + 1) * 6;
        double* b = boxes + node * 6;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            b[d] = a[d] < c[d] ? a[d] : c[d];
            b[3 + d] = a[3 + d] > c[3 + d] ? a[3 + d] : c[3 + d];
        }
    }
}

// lower bound of the squared distance from x to any point of a box, with the operation order
// of the point distance (so it never exceeds the computed distance of a contained point)
__device__ __forceinline__ double box_dist2(const double* __restrict__ b, const double x[3]) {
    double s = 0.0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        double t = 0.0;
        if (x[d] < b[d]) t = x[d] - b[d];
        else if (x[d] > b[3 + d]) t = x[d] - b[3 + d];
        s = s + t * t;
    }
    return s;  // +inf for an empty (padding) box: inf - inf never happens since lo = +inf > x
}

template <int K>
__global__ void __launch_bounds__(kBlock)
knn_query_kernel(const double* __restrict__ sorted, const int32_t* __restrict__ order, const double* __restrict__ boxes,
                 int64_t n_src, int64_t n_pow2, const double* __restrict__ tgt, int64_t n_tgt, int k,
                 int32_t* __restrict__ idx_out, double* __restrict__ d2_out) {
    const int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (t >= n_tgt) return;
    const double x[3] = {tgt[t * 3 + 0], tgt[t * 3 + 1], tgt[t * 3 + 2]};
    double best_d[K];
    int32_t best_i[K];
#pragma unroll
    for (int j = 0; j < K; ++j) { best_d[j] = INFINITY; best_i[j] = INT32_MAX; }

    int32_t stack[kStack];  // node ids < 2 * n_pow2 <= 2^29
    int sp = 0;
    stack[sp++] = 1;
    while (sp > 0) {
        const int64_t node = stack[--sp];
        const double worst = best_d[K - 1];
        if (box_dist2(boxes + node * 6, x) > worst) continue;
        if (node >= n_pow2) {
            const int64_t p0 = (node - n_pow2) * kLeaf, p1 = min(n_src, p0 + kLeaf);
            for (int64_t p = p0; p < p1; ++p) {
                const double dx = sorted[p * 3 + 0] - x[0], dy = sorted[p * 3 + 1] - x[1], dz = sorted[p * 3 + 2] - x[2];
                double s = 0.0;  // scipy: sqeuclidean_distance_double, n = 3
                s = s + dx * dx;
                s = s + dy * dy;
                s = s + dz * dz;
                const int32_t id = order[p];
                if (s < best_d[K - 1] || (s == best_d[K - 1] && id < best_i[K - 1])) {
                    // insertion into the sorted (distance, index) list
                    double cd = s;
                    int32_t ci = id;
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        const bool before = cd < best_d[j] || (cd == best_d[j] && ci < best_i[j]);
                        if (before) {
                            const double td = best_d[j]; const int32_t ti = best_i[j];
                            best_d[j] = cd; best_i[j] = ci;
                            cd = td; ci = ti;
                        }
                    }
                }
            }
        } else {
            const int64_t l = 2 * node, r = l + 1;
            const double dl = box_dist2(boxes + l * 6, x), dr = box_dist2(boxes + r * 6, x);
            // push the farther child first so the nearer one is visited next
            if (dl <= dr) {
                if (dr <= worst) stack[sp++] = (int32_t)r;
                if (dl <= worst) stack[sp++] = (int32_t)l;
            } else {
                if (dl <= worst) stack[sp++] = (int32_t)l;
                if (dr <= worst) stack[sp++] = (int32_t)r;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        if (j < k) {
            idx_out[t * k + j] = best_i[j] == INT32_MAX ? (int32_t)n_src : best_i[j];  // cKDTree: n for "not found"
            d2_out[t * k + j] = best_d[j];
        }
    }
}

static unsigned blocks_for(int64_t n) {
    int64_t b = (n + kBlock - 1) / kBlock;
    if (b > 4096) b = 4096;
    return (unsigned)(b < 1 ? 1 : b);
}

// ---- cutout mask: ray / triangle tests for every global point -------------------------------
// np.cross / np.dot of 3-vectors as the reference's per-point loop evaluates them (R: spatial.py:211-231).  np.cross is numpy
// ufuncs: products and differences, each rounded (no contraction).  np.dot of two 1-D arrays is cblas_ddot, and the x86-64
// OpenBLAS kernels numpy ships run a 3-element dot through their scalar tail loop `dot += y[i] * x[i]`, compiled with FMA:
// fma(a2, b2, fma(a1, b1, a0 * b0)).  Round 4's reference-run vectors (tests/golden/spatial_vectors.npz) caught the difference:
// two O96 points lying exactly on a triangle edge of a regular limited-area grid were classified differently by the
// two-roundings-per-term form this kernel (and the host builder) used before.
__device__ __forceinline__ void cross3(const double a[3], const double b[3], double c[3]) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ double dot3(const double a[3], const double b[3]) {
    return __builtin_fma(a[2], b[2], __builtin_fma(a[1], b[1], a[0] * b[0]));
}

__global__ void __launch_bounds__(kBlock)
cutout_inside_kernel(const double* __restrict__ g, int64_t n, const double* __restrict__ lam, int64_t n_lam,
                     const int32_t* __restrict__ nb, int k, uint8_t* __restrict__ inside) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const double d[3] = {g[i * 3], g[i * 3 + 1], g[i * 3 + 2]};
    const double epsilon = 0.0000001;
    bool hit = false;
    for (int j = 0; j < k && !hit; ++j) {
        const int64_t i0 = nb[i * k + j], i1 = nb[i * k + (j + 1) % k], i2 = nb[i * k + (j + 2) % k];
        // never read outside lam_xyz: a "not found" marker (n_lam, atx_knn_query with k > n_lam) or any other stray index
        // skips the triangle (the host wrapper rejects such tables before the launch, as the reference's indexing would)
        if (i0 < 0 || i0 >= n_lam || i1 < 0 || i1 >= n_lam || i2 < 0 || i2 >= n_lam) continue;
        double v0[3], e1[3], e2[3], s[3], h[3], q[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            v0[c] = lam[i0 * 3 + c];
            e1[c] = lam[i1 * 3 + c] - v0[c];
            e2[c] = lam[i2 * 3 + c] - v0[c];
            s[c] = 0.0 - v0[c];  // ray origin = centre of the Earth
        }
        cross3(d, e2, h);
        const double a = dot3(e1, h);
        if (-epsilon < a && a < epsilon) continue;
        const double f = 1.0 / a;
        const double u = f * dot3(s, h);
        if (u < 0.0 || u > 1.0) continue;
        cross3(s, e1, q);
        const double v = f * dot3(d, q);
        if (v < 0.0 || u + v > 1.0) continue;
        const double t = f * dot3(e2, q);
        hit = t > epsilon;
    }
    inside[i] = hit ? 1 : 0;
}

}  // namespace atx

using namespace atx;

extern "C" int atx_cutout_inside(const double* global_xyz, int64_t n, const double* lam_xyz, int64_t n_lam,
                                 const int32_t* neighbours, int32_t k, uint8_t* inside, void* stream) {
    ATX_REQUIRE(lam_xyz && ((global_xyz && neighbours && inside) || n == 0), ATX_EINVAL, "atx_cutout_inside: null pointer");
    ATX_REQUIRE(n >= 0 && n_lam > 0 && k >= 1 && k <= kMaxK, ATX_EINVAL, "atx_cutout_inside: bad sizes (n=%lld, n_lam=%lld, k=%d)",
                (long long)n, (long long)n_lam, k);
    if (n == 0) return ATX_OK;
    hipLaunchKernelGGL(cutout_inside_kernel, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0,
                       static_cast<hipStream_t>(stream), global_xyz, n, lam_xyz, n_lam, neighbours, k, inside);
    ATX_LAUNCH_CHECK("cutout_inside");
    return ATX_OK;
}

extern "C" size_t atx_knn_workspace_bytes(int64_t n_src) {
    if (n_src <= 0 || n_src > INT32_MAX) return 0;
    return knn_layout(n_src).total_bytes;
}

extern "C" int atx_knn_build(const double* src_xyz, int64_t n_src, void* workspace, size_t workspace_bytes, void* stream) {
    ATX_REQUIRE(src_xyz && workspace, ATX_EINVAL, "atx_knn_build: null pointer");
    ATX_REQUIRE(n_src > 0 && n_src <= INT32_MAX, ATX_EINVAL, "atx_knn_build: n_src=%lld outside (0, 2^31)", (long long)n_src);
    const KnnHeader h = knn_layout(n_src);
    ATX_REQUIRE(workspace_bytes >= h.total_bytes, ATX_EWORKSPACE, "atx_knn_build: workspace %zu < %zu", workspace_bytes, h.total_bytes);
    ATX_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, ATX_EALIGN, "atx_knn_build: workspace must be 256-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    char* base = static_cast<char*>(workspace);
    double* sorted = reinterpret_cast<double*>(base + h.off_xyz);
    int32_t* order = reinterpret_cast<int32_t*>(base + h.off_order);
    double* boxes = reinterpret_cast<double*>(base + h.off_boxes);
    uint64_t* keys_in = reinterpret_cast<uint64_t*>(base + h.off_keys_in);
    uint64_t* keys_out = reinterpret_cast<uint64_t*>(base + h.off_keys_out);
    int32_t* vals_in = reinterpret_cast<int32_t*>(base + h.off_vals_in);

    int st = hip_status(hipMemcpyAsync(base, &h, sizeof(KnnHeader), hipMemcpyHostToDevice, s), "atx_knn_build header");
    if (st != ATX_OK) return st;
    hipLaunchKernelGGL(knn_keys_kernel, dim3(blocks_for(n_src)), dim3(kBlock), 0, s, src_xyz, n_src, keys_in, vals_in);
    ATX_LAUNCH_CHECK("knn_keys");
    size_t tmp = h.tmp_bytes;
    st = hip_status(hipcub::DeviceRadixSort::SortPairs(base + h.off_tmp, tmp, keys_in, keys_out, vals_in, order, (int)n_src, 0, 63, s),
                    "atx_knn_build sort");
    if (st != ATX_OK) return st;
    hipLaunchKernelGGL(knn_gather_kernel, dim3(blocks_for(n_src)), dim3(kBlock), 0, s, src_xyz, order, n_src, sorted);
    ATX_LAUNCH_CHECK("knn_gather");
    hipLaunchKernelGGL(knn_leaf_boxes_kernel, dim3(blocks_for(h.n_pow2)), dim3(kBlock), 0, s, sorted, n_src, h.n_pow2, boxes);
    ATX_LAUNCH_CHECK("knn_leaf_boxes");
    for (int64_t count = h.n_pow2 / 2; count >= 1; count /= 2) {
        hipLaunchKernelGGL(knn_merge_boxes_kernel, dim3(blocks_for(count)), dim3(kBlock), 0, s, boxes, count, count);
        ATX_LAUNCH_CHECK("knn_merge_boxes");
    }
    return ATX_OK;
}

extern "C" int atx_knn_query(const void* workspace, int64_t n_src, const double* tgt_xyz, int64_t n_tgt, int32_t k,
                             int32_t* idx_out, double* d2_out, void* stream) {
    ATX_REQUIRE(workspace && tgt_xyz && idx_out && d2_out, ATX_EINVAL, "atx_knn_query: null pointer");
    ATX_REQUIRE(n_src > 0 && n_src <= INT32_MAX && n_tgt >= 0, ATX_EINVAL, "atx_knn_query: bad sizes");
    ATX_REQUIRE(k >= 1 && k <= kMaxK, ATX_EINVAL, "atx_knn_query: k=%d outside [1, %d]", k, kMaxK);
    if (n_tgt == 0) return ATX_OK;
    const KnnHeader h = knn_layout(n_src);  // same layout as the build
    hipStream_t s = static_cast<hipStream_t>(stream);
    const char* base = static_cast<const char*>(workspace);
    const double* sorted = reinterpret_cast<const double*>(base + h.off_xyz);
    const int32_t* order = reinterpret_cast<const int32_t*>(base + h.off_order);
    const double* boxes = reinterpret_cast<const double*>(base + h.off_boxes);
    const unsigned grid = (unsigned)((n_tgt + kBlock - 1) / kBlock);
#define ATX_KNN_LAUNCH(KK)                                                                                               \
    hipLaunchKernelGGL((knn_query_kernel<KK>), dim3(grid), dim3(kBlock), 0, s, sorted, order, boxes, n_src, h.n_pow2,   \
                       tgt_xyz, n_tgt, k, idx_out, d2_out)
    if (k == 1) ATX_KNN_LAUNCH(1);
    else if (k <= 4) {
        // K must equal k: the pruning bound is the K-th best
        if (k == 2) ATX_KNN_LAUNCH(2);
        else if (k == 3) ATX_KNN_LAUNCH(3);
        else ATX_KNN_LAUNCH(4);
    } else if (k <= 8) {
        if (k == 5) ATX_KNN_LAUNCH(5);
        else if (k == 6) ATX_KNN_LAUNCH(6);
        else if (k == 7) ATX_KNN_LAUNCH(7);
        else ATX_KNN_LAUNCH(8);
    } else {
        switch (k) {
            case 9: ATX_KNN_LAUNCH(9); break;
            case 10: ATX_KNN_LAUNCH(10); break;
            case 11: ATX_KNN_LAUNCH(11); break;
            case 12: ATX_KNN_LAUNCH(12); break;
            case 13: ATX_KNN_LAUNCH(13); break;
            case 14: ATX_KNN_LAUNCH(14); break;
            case 15: ATX_KNN_LAUNCH(15); break;
            case 16: ATX_KNN_LAUNCH(16); break;
            default: ATX_KNN_LAUNCH(17); break;
        }
    }
#undef ATX_KNN_LAUNCH
    ATX_LAUNCH_CHECK("knn_query");
    return ATX_OK;
}
