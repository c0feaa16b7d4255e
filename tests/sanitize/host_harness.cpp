// Sanitizer harness for the host C++ of the batched pipeline (tests/test_sanitizers.py builds it with -fsanitize=thread and with
// -fsanitize=address,undefined; no GPU, no HIP runtime): four caller threads -- the bench's worker groups -- each push 384 streams
// through ddk::parallel_for for a few hundred steps; a stream's body runs the phases csrc/pipeline.hip runs on the pool: box hygiene
// (host_phases.h clean_boxes), a cascade-sized assignment problem (lsap.cpp), the CPython-set iteration order of the IoU stage's
// candidates (pyset.cpp) and the count-line segment test (host_phases.h seg_intersect).  Every stream's results are checked
// against a serial replay, so a lost or doubled chunk shows as a mismatch as well as a race report.
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
#include "../../deepdish_amd/csrc/hostpool.h"
#include "../../deepdish_amd/csrc/host_phases.h"

void dd_set_error(const char *, ...) {}
namespace ddk {
int lsap(const double *cost, int nr, int nc, int *rows, int *cols);
void pyset_difference_order(const std::vector<int> &a, const std::vector<int> &b, std::vector<int> &out);
}

struct Rng { unsigned long long s; unsigned next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(s >> 33); } double uni() { return next() / 2147483648.0; } };

struct Result { long long boxes, assigned, order_sum, crossings; bool operator==(const Result &o) const { return boxes == o.boxes && assigned == o.assigned && order_sum == o.order_sum && crossings == o.crossings; } };

static Result one_stream(int group, int z, int step) {
    Rng r{(unsigned long long)(group * 1000003 + z * 7919 + step * 104729 + 1)};
    Result out{0, 0, 0, 0};
    const int k = 5 + r.next() % 20;
    std::vector<double> boxes, scores; std::vector<int> cls;
    for (int i = 0; i < k; ++i) {
        boxes.insert(boxes.end(), {r.uni() * 700 - 30, r.uni() * 520 - 20, r.uni() * 120, r.uni() * 200});
        scores.push_back(r.uni()); cls.push_back(r.next() % 3);
    }
    if (r.next() % 50 == 0) boxes[1] = std::nan("");
    std::vector<int64_t> ib; std::vector<double> is; std::vector<int> ic;
    ddhost::clean_boxes(boxes, scores, cls, 640, 480, ib, is, ic);
    out.boxes = (long long)is.size();
    const int nr = 3 + r.next() % 14, nc = (int)is.size();
    if (nc > 0) {
        std::vector<double> cost((size_t)nr * nc);
        for (double &c : cost) c = r.uni() < 0.3 ? 0.20001 : r.uni() * 0.2;
        std::vector<int> rows(std::min(nr, nc)), cols(std::min(nr, nc));
        const int n = ddk::lsap(cost.data(), nr, nc, rows.data(), cols.data());
        for (int i = 0; i < n; ++i) out.assigned += rows[i] * 31 + cols[i];
    }
    std::vector<int> a, b, ord;
    for (int i = 0; i < nr + 10; ++i) a.push_back(r.next() % 200);
    for (int i = 0; i < 6; ++i) b.push_back(r.next() % 200);
    ddk::pyset_difference_order(a, b, ord);
    for (size_t i = 0; i < ord.size(); ++i) out.order_sum += (long long)(i + 1) * ord[i];
    const double line[4] = {320, 0, 320, 480};
    for (int i = 0; i < 8; ++i) {
        const double p[2] = {r.uni() * 640, r.uni() * 480}, q[2] = {r.uni() * 640, r.uni() * 480};
        out.crossings += ddhost::seg_intersect(line, line + 2, p, q);
    }
    return out;
}

int main(int argc, char **argv) {
    const int groups = 4, S = 384, steps = argc > 1 ? atoi(argv[1]) : 300;
    std::atomic<long long> bad{0};
    std::vector<std::thread> th;
    for (int g = 0; g < groups; ++g)
        th.emplace_back([&, g] {
            std::vector<Result> res(S);
            for (int step = 0; step < steps; ++step) {
                ddk::parallel_for(S, 8, [&](int z0, int z1) { for (int z = z0; z < z1; ++z) res[z] = one_stream(g, z, step); });
                if (step % 37 == 0)
                    for (int z = 0; z < S; ++z) if (!(res[z] == one_stream(g, z, step))) bad++;
            }
        });
    for (auto &t : th) t.join();
    printf("host harness: %d groups x %d streams x %d steps on %d pool threads, %lld mismatches\n", groups, S, steps, ddk::host_threads(), bad.load());
    return bad.load() ? 1 : 0;
}
