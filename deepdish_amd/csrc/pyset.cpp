// Iteration order of `list(set(A) - set(B))` for small non-negative ints, as CPython 3.10 produces it.
//
// deep_sort builds the IoU-stage candidate list from exactly that expression
// (deep_sort/linear_assignment.py:140 `list(set(track_indices) - set(k for k, _ in matches))`,
// consumed at deep_sort/tracker.py:120-123), so the ROW ORDER of the IoU assignment problem is CPython's
// set iteration order.  The order only matters through exact ties among clamped costs, but then it
// decides the order of `unmatched_detections` and hence of new track ids -- so it is reproduced here
// rather than approximated.  Model (Objects/setobject.c, CPython 3.10): open addressing, table sizes
// 8 << n, hash(int) = value, LINEAR_PROBES = 9, PERTURB_SHIFT = 5, resize when fill * 5 >= mask * 3 to
// the first size > used * 4; `a - b` copies `a` and discards when len(a) / 4 > len(b), else re-inserts
// the survivors of `a` (in a's table order) into a fresh set.  tests/test_pyset.py checks it against
// the running interpreter on random inputs.
#include <cstddef>
#include <vector>
#include "common.h"

namespace {

constexpr int LINEAR_PROBES = 9;
constexpr int PERTURB_SHIFT = 5;
constexpr long long EMPTY = -1, DUMMY = -2;

struct PySet {
    std::vector<long long> tab;          // key, EMPTY or DUMMY
    size_t mask = 7, fill = 0, used = 0;
    PySet() : tab(8, EMPTY) {}

    static void insert_clean(std::vector<long long> &t, size_t mask, long long key) {
        size_t perturb = (size_t)key, i = (size_t)key & mask;
        for (;;) {
            if (t[i] == EMPTY) { t[i] = key; return; }
            if (i + LINEAR_PROBES <= mask)
                for (int j = 1; j <= LINEAR_PROBES; ++j)
                    if (t[i + j] == EMPTY) { t[i + j] = key; return; }
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }

    void resize(size_t minused) {
        size_t newsize = 8;
        while (newsize <= minused) newsize <<= 1;
        std::vector<long long> nt(newsize, EMPTY);
        for (long long k : tab)
            if (k >= 0) insert_clean(nt, newsize - 1, k);
        tab.swap(nt);
        mask = newsize - 1;
        fill = used;
    }

    bool contains(long long key) const {
        size_t perturb = (size_t)key, i = (size_t)key & mask;
        for (;;) {
            int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
            size_t e = i;
            do {
                if (tab[e] == EMPTY) return false;
                if (tab[e] == key) return true;
                ++e;
            } while (probes--);
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }

    void add(long long key) {                         // set_add_entry
        size_t perturb = (size_t)key, i = (size_t)key & mask;
        long long freeslot = -1;
        for (;;) {
            int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
            size_t e = i;
            do {
                if (tab[e] == EMPTY) {
                    if (freeslot >= 0) { tab[(size_t)freeslot] = key; ++used; return; }
                    tab[e] = key;
                    ++fill; ++used;
                    if (fill * 5 >= mask * 3) resize(used > 50000 ? used * 2 : used * 4);
                    return;
                }
                if (tab[e] == key) return;
                if (tab[e] == DUMMY) freeslot = (long long)e;
                ++e;
            } while (probes--);
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }

    void discard(long long key) {                     // set_discard_entry: leaves a dummy
        size_t perturb = (size_t)key, i = (size_t)key & mask;
        for (;;) {
            int probes = (i + LINEAR_PROBES <= mask) ? LINEAR_PROBES : 0;
            size_t e = i;
            do {
                if (tab[e] == EMPTY) return;
                if (tab[e] == key) { tab[e] = DUMMY; --used; return; }
                ++e;
            } while (probes--);
            perturb >>= PERTURB_SHIFT;
            i = (i * 5 + 1 + perturb) & mask;
        }
    }

    // set_merge into an EMPTY set (what set_copy does)
    void merge_from(const PySet &o) {
        if (o.used == 0) return;
        if ((fill + o.used) * 5 >= mask * 3) resize((used + o.used) * 2);
        if (fill == 0 && mask == o.mask && o.fill == o.used) { tab = o.tab; fill = o.fill; used = o.used; return; }
        fill = o.used; used = o.used;
        for (long long k : o.tab)
            if (k >= 0) insert_clean(tab, mask, k);
    }
};

}  // namespace

namespace ddk {

// out = list(set(a) - set(b)) in CPython 3.10 iteration order.  a, b: non-negative ints.
void pyset_difference_order(const std::vector<int> &a, const std::vector<int> &b, std::vector<int> &out) {
    PySet sa, sb;
    for (int v : a) sa.add(v);
    for (int v : b) sb.add(v);
    out.clear();
    if ((sa.used >> 2) > sb.used) {                   // set_copy_and_difference
        PySet r;
        r.merge_from(sa);
        for (long long k : sb.tab)
            if (k >= 0) r.discard(k);
        for (long long k : r.tab)
            if (k >= 0) out.push_back((int)k);
        return;
    }
    PySet r;
    for (long long k : sa.tab)
        if (k >= 0 && !sb.contains(k)) r.add(k);
    for (long long k : r.tab)
        if (k >= 0) out.push_back((int)k);
}

}  // namespace ddk

// Test hook (host only): out_host receives the ordered difference, *out_n its length.
extern "C" int dd_pyset_difference_order_host(const int *a_host, int na, const int *b_host, int nb, int *out_host, int *out_n_host) {
    DD_REQUIRE(na >= 0 && nb >= 0 && out_n_host && (na == 0 || (a_host && out_host)) && (nb == 0 || b_host), DD_E_ARG,
               "dd_pyset_difference_order_host: bad argument");
    std::vector<int> a(a_host, a_host + na), b(b_host, b_host + nb), o;
    for (int v : a) DD_REQUIRE(v >= 0, DD_E_ARG, "dd_pyset_difference_order_host: negative element");
    for (int v : b) DD_REQUIRE(v >= 0, DD_E_ARG, "dd_pyset_difference_order_host: negative element");
    ddk::pyset_difference_order(a, b, o);
    for (size_t i = 0; i < o.size(); ++i) out_host[i] = o[i];
    *out_n_host = (int)o.size();
    return DD_OK;
}
