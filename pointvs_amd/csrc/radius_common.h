// Exact cdist-style distance decisions shared by radius_graph.hip and screen_graph.hip.
#pragma once
#include "common.h"

namespace {

// Distances: scipy euclidean_distance_double: s = 0; s += d*d for k = 0,1,2; d = sqrt(s)  (fp64, no FMA).
// `sqrt(s) < r` and `sqrt(s) > 1e-7` exactly as the reference decides them, with the correctly
// rounded square root only evaluated in the (practically never taken) band where comparing s with
// r*r could disagree with it.
struct Radius {
    double r, lo, hi;   // s < lo => sqrt(s) < r for sure; s > hi => sqrt(s) >= r for sure
};
__host__ __device__ inline Radius make_radius(double r) {
    Radius q;
    q.r = r;
    q.lo = r * r * (1.0 - 0x1p-48);
    q.hi = r * r * (1.0 + 0x1p-48);
    return q;
}
__device__ __forceinline__ bool below(double s, const Radius& q) {
    if (s < q.lo) return true;
    if (s > q.hi) return false;
    return __dsqrt_rn(s) < q.r;
}
__device__ __forceinline__ bool above(double s, const Radius& q) {
    if (s > q.hi) return true;
    if (s < q.lo) return false;
    return __dsqrt_rn(s) > q.r;
}


// squared distance as scipy's euclidean_distance_double accumulates it (fp64, no FMA contraction)
__device__ __forceinline__ double pvs_sqdist(double xi, double yi, double zi, double xj, double yj, double zj) {
    const double d0 = xi - xj, d1 = yi - yj, d2 = zi - zj;
    double s = __dmul_rn(d0, d0);
    s = __dadd_rn(s, __dmul_rn(d1, d1));
    s = __dadd_rn(s, __dmul_rn(d2, d2));
    return s;
}

}  // namespace
