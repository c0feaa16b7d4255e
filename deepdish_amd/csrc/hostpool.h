// Process-wide pool of host threads for the per-stream host phases of the batched pipeline (detector adaptor filter,
// box hygiene, matching cascade + LSAP, count line: deepdish.py:940-960,1028-1114 run once per stream and the streams
// share nothing).  The reference runs one Python process per camera; here S streams share one process, so their host
// phases are spread over a few threads instead of being walked by one.
//
//   ddk::parallel_for(n, grain, [&](int i0, int i1) { ... });   // [i0, i1) chunks of `grain` items, any thread, any order
//
// The calling thread takes chunks too and returns when every chunk has run.  Several callers (worker groups) may be
// inside parallel_for at once; they share the pool.  The body must not call into HIP (pool threads carry no device
// context) and must not throw.  DD_HOST_THREADS = pool threads per process (0 = run everything on the caller;
// default min(8, cores / 2) -- a one-GPU box gives a rank 16 cores, bench.py sets it per rank for N ranks).
#pragma once
#include <functional>

namespace ddk {
void parallel_for(int n, int grain, const std::function<void(int, int)> &body);
int host_threads();
}  // namespace ddk
