// Rectangular linear sum assignment on the host (shortest augmenting paths with dual
// variables; D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES 2016).
//
// Replaces scipy.optimize.linear_sum_assignment as called at deep_sort/linear_assignment.py:58.
// scipy is a third-party dependency of the reference (not vendored; 1.15.3 in the build image).
// The optimum is unique whenever the costs are tie-free, but deep_sort clamps every infeasible
// entry to max_distance + 1e-5 (linear_assignment.py:57), so ties are the norm -- and WHICH of the
// clamped pairs gets picked decides the order of `unmatched_detections`, hence the order in which
// new track ids are handed out (tracker.py:78-79).  To keep track ids bit-identical this solver
// therefore fixes the same deterministic choices as that solver: rows are inserted in index order,
// the candidate-column list starts in descending column order and shrinks by swap-with-last,
// the scan keeps the first strictly smaller reduced cost and, on an exact tie, prefers a column
// that is still free.  tests/test_lsap.py checks agreement on tie-heavy matrices.
#include <vector>
#include <limits>
#include <algorithm>
#include <numeric>
#include "common.h"

namespace {

struct Solver {
    int nr, nc;                       // nr <= nc
    const double *c;                  // row-major nr x nc
    std::vector<double> u, v, dist;
    std::vector<int> pred, col_of_row, row_of_col, cand;
    std::vector<char> row_seen, col_seen;

    Solver(int nr_, int nc_, const double *c_)
        : nr(nr_), nc(nc_), c(c_), u(nr_, 0.0), v(nc_, 0.0), dist(nc_), pred(nc_, -1),
          col_of_row(nr_, -1), row_of_col(nc_, -1), cand(nc_), row_seen(nr_), col_seen(nc_) {}

    // Dijkstra over the reduced costs from `start`; returns the free column reached, or -1.
    int search(int start, double *reach) {
        double base = 0.0;
        int live = nc;
        for (int t = 0; t < nc; ++t) cand[t] = nc - 1 - t;
        std::fill(row_seen.begin(), row_seen.end(), 0);
        std::fill(col_seen.begin(), col_seen.end(), 0);
        std::fill(dist.begin(), dist.end(), std::numeric_limits<double>::infinity());
        int row = start, sink = -1;
        while (sink < 0) {
            int pick = -1;
            double low = std::numeric_limits<double>::infinity();
            row_seen[row] = 1;
            const double *crow = c + (size_t)row * nc;
            for (int t = 0; t < live; ++t) {
                const int j = cand[t];
                const double r = base + crow[j] - u[row] - v[j];
                if (r < dist[j]) { dist[j] = r; pred[j] = row; }
                if (dist[j] < low || (dist[j] == low && row_of_col[j] < 0)) { low = dist[j]; pick = t; }
            }
            base = low;
            if (!(base < std::numeric_limits<double>::infinity())) return -1;
            const int j = cand[pick];
            if (row_of_col[j] < 0) sink = j; else row = row_of_col[j];
            col_seen[j] = 1;
            cand[pick] = cand[--live];
        }
        *reach = base;
        return sink;
    }

    bool run() {
        for (int cur = 0; cur < nr; ++cur) {
            double reach = 0.0;
            const int sink = search(cur, &reach);
            if (sink < 0) return false;
            u[cur] += reach;
            for (int i = 0; i < nr; ++i)
                if (row_seen[i] && i != cur) u[i] += reach - dist[col_of_row[i]];
            for (int j = 0; j < nc; ++j)
                if (col_seen[j]) v[j] -= reach - dist[j];
            int j = sink;
            for (;;) {                                   // flip the alternating path
                const int i = pred[j];
                row_of_col[j] = i;
                std::swap(col_of_row[i], j);
                if (i == cur) break;
            }
        }
        return true;
    }
};

}  // namespace

namespace ddk {

// cost row-major [nr][nc]; writes min(nr,nc) pairs sorted by row. Returns pair count or -1.
int lsap(const double *cost, int nr, int nc, int *rows, int *cols) {
    if (nr <= 0 || nc <= 0) return 0;
    for (size_t i = 0, n = (size_t)nr * nc; i < n; ++i)
        if (cost[i] != cost[i] || cost[i] == -std::numeric_limits<double>::infinity()) return -1;
    if (nr <= nc) {
        Solver s(nr, nc, cost);
        if (!s.run()) return -1;
        for (int i = 0; i < nr; ++i) { rows[i] = i; cols[i] = s.col_of_row[i]; }
        return nr;
    }
    std::vector<double> tr((size_t)nr * nc);               // tall: solve the transpose
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < nc; ++j) tr[(size_t)j * nr + i] = cost[(size_t)i * nc + j];
    Solver s(nc, nr, tr.data());
    if (!s.run()) return -1;
    std::vector<int> order(nc);
    std::iota(order.begin(), order.end(), 0);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return s.col_of_row[a] < s.col_of_row[b]; });
    for (int t = 0; t < nc; ++t) { rows[t] = s.col_of_row[order[t]]; cols[t] = order[t]; }
    return nc;
}

}  // namespace ddk

extern "C" int dd_lsap_host(const double *cost_host, int nr, int nc, int *row_ind_host, int *col_ind_host) {
    DD_REQUIRE(nr >= 0 && nc >= 0, DD_E_ARG, "dd_lsap_host: negative shape");
    if (nr == 0 || nc == 0) return DD_OK;
    DD_REQUIRE(cost_host && row_ind_host && col_ind_host, DD_E_ARG, "dd_lsap_host: NULL argument");
    const int n = ddk::lsap(cost_host, nr, nc, row_ind_host, col_ind_host);
    DD_REQUIRE(n >= 0, DD_E_ARG, "dd_lsap_host: cost matrix is infeasible or contains NaN/-inf");
    return DD_OK;
}
