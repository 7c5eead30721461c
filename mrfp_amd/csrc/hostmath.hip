// hostmath.hip -- host-side numerics of the whitening-loss setup (no kernels).
//
// mrfp_kmeans1d: globally optimal k-means of scalars.  The reference selects the "sensitive" covariance entries of the
// ISW loss with `kmeans1d.cluster(var_flatten, 50)` (reference network/cov_settings.py:57-59; kmeans1d is a PyPI
// dependency that is NOT vendored in the reference tree: its published algorithm is the dynamic programme over the sorted
// values, D[q][i] = min_j D[q-1][j-1] + cost(j, i), with cost = within-cluster sum of squares from prefix sums and the
// row minima found through the monotonicity of the optimal split).  Restated here with the divide-and-conquer form of
// that monotone search, O(k n log n): 50 clusters over the 65 536 entries of a 256 x 256 variance matrix take ~0.3 s.
// Labels come back in the order of the input, clusters numbered by ascending centroid (as kmeans1d does), so
// `clusters.count(0)` of cov_settings.py:58 is the size of the lowest cluster.
#include <algorithm>
#include <numeric>
#include <vector>

#include "common.hpp"

namespace mrfp {
namespace {

struct KmCtx {
    const std::vector<double>* ps;    // prefix sums of the sorted values
    const std::vector<double>* ps2;   // prefix sums of their squares
    // within-cluster sum of squares of sorted[j .. i] (inclusive), via the mean of the segment
    double cost(int64_t j, int64_t i) const {
        const double s = (*ps)[i + 1] - (*ps)[j], q = (*ps2)[i + 1] - (*ps2)[j];
        const double c = q - s * s / (double)(i - j + 1);
        return c > 0.0 ? c : 0.0;
    }
};

// fill cur[lo..hi] given prev (the row for one cluster fewer); the optimal first index of the last cluster is
// non-decreasing in i, so the midpoint's optimum splits the candidate range
void km_row(const KmCtx& c, const std::vector<double>& prev, std::vector<double>& cur, std::vector<int32_t>& arg, int q,
            int64_t lo, int64_t hi, int64_t jlo, int64_t jhi) {
    if (lo > hi) return;
    const int64_t mid = (lo + hi) >> 1;
    double best = 0.0;
    int64_t bj = -1;
    const int64_t jend = std::min(mid, jhi);
    for (int64_t j = std::max<int64_t>(jlo, q); j <= jend; ++j) {      // the last cluster is sorted[j .. mid], q clusters before it
        const double v = prev[j - 1] + c.cost(j, mid);
        if (bj < 0 || v < best) { best = v; bj = j; }
    }
    cur[mid] = best;
    arg[mid] = (int32_t)bj;
    km_row(c, prev, cur, arg, q, lo, mid - 1, jlo, bj);
    km_row(c, prev, cur, arg, q, mid + 1, hi, bj, jhi);
}

}  // namespace
}  // namespace mrfp

extern "C" int mrfp_kmeans1d(const double* x, int64_t n, int k, int32_t* labels, double* centroids) {
    using namespace mrfp;
    MRFP_CHECK(x && labels && centroids && n > 0 && k > 0 && n < (int64_t)1 << 31, "kmeans1d: bad arguments");
    if ((int64_t)k > n) k = (int)n;
    std::vector<int64_t> order((size_t)n);
    std::iota(order.begin(), order.end(), (int64_t)0);
    std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return x[a] < x[b]; });
    std::vector<double> ps((size_t)n + 1, 0.0), ps2((size_t)n + 1, 0.0);
    // sums about the median keep the prefix differences well conditioned
    const double shift = x[order[(size_t)n / 2]];
    for (int64_t i = 0; i < n; ++i) {
        const double v = x[order[(size_t)i]] - shift;
        ps[(size_t)i + 1] = ps[(size_t)i] + v;
        ps2[(size_t)i + 1] = ps2[(size_t)i] + v * v;
    }
    KmCtx c{&ps, &ps2};
    std::vector<double> prev((size_t)n), cur((size_t)n);
    std::vector<std::vector<int32_t>> arg((size_t)k, std::vector<int32_t>((size_t)n, 0));
    for (int64_t i = 0; i < n; ++i) prev[(size_t)i] = c.cost(0, i);
    for (int q = 1; q < k; ++q) {
        // with q + 1 clusters the first q points cannot be covered: start the row at i = q
        for (int64_t i = 0; i < q; ++i) cur[(size_t)i] = 0.0;
        km_row(c, prev, cur, arg[(size_t)q], q, q, n - 1, q, n - 1);
        prev.swap(cur);
    }
    // backtrack
    int64_t hi = n - 1;
    for (int q = k - 1; q >= 0; --q) {
        const int64_t lo = q == 0 ? 0 : (int64_t)arg[(size_t)q][(size_t)hi];
        const double s = ps[(size_t)hi + 1] - ps[(size_t)lo];
        centroids[q] = s / (double)(hi - lo + 1) + shift;
        for (int64_t i = lo; i <= hi; ++i) labels[order[(size_t)i]] = q;
        hi = lo - 1;
    }
    return 0;
}
