// stream_handover.hip -- what a hand-over between the compute stream and a side stream costs per PCG iteration
// (the two-phase data-parallel product: G_a | side: all-reduce(tail) || G_b | wait | K1-K3), with stand-in kernels.
//
//   hipcc --offload-arch=gfx950 -O2 -o stream_handover stream_handover.hip && ./stream_handover
//
// Three hipGraphs of busy-wait kernels stand for G_a (44 launches), G_b (26) and K1-K3 (3); a 20 us kernel on the
// side stream stands for the tail's all-reduce.  Variants of the hand-over, one line each (us per iteration):
//   none        no side stream at all (the floor: three graph launches back to back)
//   same        the stand-in collective on the compute stream (no overlap, no hand-over)
//   events      hipEventRecord / hipStreamWaitEvent both ways (what session.reduce_phases does)
//   values      hipStreamWriteValue32 / hipStreamWaitValue32 on signal memory both ways
//   flag        the last node of G_a stores the iteration number; a polling kernel on the side stream waits for it;
//               the way back likewise (first node of the K1-K3 graph polls)
//   events_fwd  events for compute -> side only, nothing waits for the side stream (wrong; isolates the way back)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(e)                                                                     \
  do {                                                                            \
    hipError_t _e = (e);                                                          \
    if (_e != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(_e));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__global__ void busy(long long ticks, float* sink) {
  const long long t0 = wall_clock64();
  float x = threadIdx.x;
  while (wall_clock64() - t0 < ticks) x = x * 1.0001f + 1.f;
  if (x == 12345.f) *sink = x;
}

__global__ void store_flag(unsigned* flag, const unsigned* it) {
  if (threadIdx.x == 0) __hip_atomic_store(flag, *it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void bump(unsigned* it) { *it += 1; }
// bounded: gives up after ~50 ms and reports it (a missed hand-over must be visible, not silent)
__global__ void poll_flag(const unsigned* flag, const unsigned* it, unsigned* timeouts) {
  if (threadIdx.x != 0) return;
  const unsigned want = *it;
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) {
    __builtin_amdgcn_s_sleep(8);
    if (wall_clock64() - t0 > 5000000LL) { atomicAdd(timeouts, 1u); break; }
  }
}

static long long ticks_per_us() {
  int khz = 0;
  CK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, 0));
  return khz / 1000;
}

struct Graph {
  hipGraph_t g = nullptr;
  hipGraphExec_t e = nullptr;
};

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 400;
  CK(hipSetDevice(0));
  hipStream_t s, side;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
  const long long us = ticks_per_us();
  float* sink;
  CK(hipMalloc(&sink, 4));
  unsigned *flag_a, *flag_t, *it_a, *it_t, *timeouts;
  CK(hipMalloc(&flag_a, 4)); CK(hipMalloc(&flag_t, 4)); CK(hipMalloc(&it_a, 4)); CK(hipMalloc(&it_t, 4));
  CK(hipMalloc(&timeouts, 4));
  unsigned long long *sig_a = nullptr, *sig_t = nullptr;
  int can_wait = 0;
  CK(hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0));
  if (can_wait) {
    if (hipExtMallocWithFlags((void**)&sig_a, 8, hipMallocSignalMemory) != hipSuccess ||
        hipExtMallocWithFlags((void**)&sig_t, 8, hipMallocSignalMemory) != hipSuccess)
      can_wait = 0;
  }
  printf("{\"note\": \"ticks/us %lld, stream wait-value support %d\"}\n", us, can_wait);

  auto capture = [&](int n, long long each_us, bool tail_flag, bool head_poll) {
    Graph G;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    if (head_poll) { bump<<<1, 1, 0, s>>>(it_t); poll_flag<<<1, 64, 0, s>>>(flag_t, it_t, timeouts); }
    for (int i = 0; i < n; ++i) busy<<<256, 256, 0, s>>>(each_us * us, sink);
    if (tail_flag) { bump<<<1, 1, 0, s>>>(it_a); store_flag<<<1, 64, 0, s>>>(flag_a, it_a); }
    CK(hipStreamEndCapture(s, &G.g));
    CK(hipGraphInstantiate(&G.e, G.g, nullptr, nullptr, 0));
    return G;
  };
  Graph ga = capture(44, 8, false, false), gb = capture(26, 8, false, false), gk = capture(3, 25, false, false);
  Graph ga_f = capture(44, 8, true, false), gk_p = capture(3, 25, false, true);

  hipEvent_t ev_a, ev_t, lagged[4];
  CK(hipEventCreateWithFlags(&ev_a, hipEventDisableTiming));
  CK(hipEventCreateWithFlags(&ev_t, hipEventDisableTiming));
  for (auto& e : lagged) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  unsigned *side_it;  // iteration counter the side stream's kernels read
  CK(hipMalloc(&side_it, 4));

  if (argc > 2 && !strcmp(argv[2], "queues")) {
    // which side streams run beside the compute stream, and what the hand-over by stream values costs on each
    for (int null_compute = 0; null_compute < 2; ++null_compute) {
      hipStream_t cs = null_compute ? nullptr : s;
      hipStream_t cand[10];
      for (auto& c : cand) CK(hipStreamCreateWithFlags(&c, hipStreamNonBlocking));
      for (int k = 0; k < 10; ++k) {
        // probe: 300 us on the compute stream, 5 us on the candidate; concurrent iff the short one ends first
        hipEvent_t e_long, e_short;
        CK(hipEventCreate(&e_long)); CK(hipEventCreate(&e_short));
        CK(hipDeviceSynchronize());
        busy<<<256, 256, 0, cs>>>(300 * us, sink);
        CK(hipEventRecord(e_long, cs));
        busy<<<1, 64, 0, cand[k]>>>(5 * us, sink);
        CK(hipEventRecord(e_short, cand[k]));
        CK(hipEventSynchronize(e_short));
        const int beside = hipEventQuery(e_long) == hipErrorNotReady;
        CK(hipDeviceSynchronize());
        double best = 1e30;
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipMemset(sig_a, 0, 8)); CK(hipMemset(sig_t, 0, 8));
          CK(hipDeviceSynchronize());
          auto t0 = std::chrono::steady_clock::now();
          for (int i = 1; i <= iters; ++i) {
            const unsigned tag = (unsigned)i;
            CK(hipGraphLaunch(ga.e, cs));
            CK(hipStreamWriteValue64(cs, sig_a, tag, 0));
            CK(hipStreamWaitValue64(cand[k], sig_a, tag, hipStreamWaitValueGte, ~0ull));
            busy<<<64, 256, 0, cand[k]>>>(20 * us, sink);
            CK(hipStreamWriteValue64(cand[k], sig_t, tag, 0));
            CK(hipGraphLaunch(gb.e, cs));
            CK(hipStreamWaitValue64(cs, sig_t, tag, hipStreamWaitValueGte, ~0ull));
            CK(hipGraphLaunch(gk.e, cs));
            CK(hipEventRecord(lagged[i & 3], cs));
            if (i > 1) CK(hipEventSynchronize(lagged[(i - 1) & 3]));
          }
          CK(hipDeviceSynchronize());
          const double per = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
          if (per < best) best = per;
        }
        printf("{\"compute_stream\": \"%s\", \"side_stream\": %d, \"probe_runs_beside\": %d, \"values_us_per_iteration\": %.1f}\n",
               null_compute ? "null" : "created", k, beside, best);
        fflush(stdout);
      }
    }
    return 0;
  }
  const char* names[] = {"none", "same", "events", "values", "flag", "events_fwd", "none", "values64"};
  for (int v = 0; v < 8; ++v) {
    if ((v == 3 || v == 7) && !can_wait) { printf("{\"variant\": \"values\", \"skipped\": \"no stream wait-value support\"}\n"); continue; }
    CK(hipMemset(flag_a, 0, 4)); CK(hipMemset(flag_t, 0, 4)); CK(hipMemset(it_a, 0, 4)); CK(hipMemset(it_t, 0, 4));
    CK(hipMemset(side_it, 0, 4)); CK(hipMemset(timeouts, 0, 4));
    if (can_wait) { CK(hipMemset(sig_a, 0, 8)); CK(hipMemset(sig_t, 0, 8)); }
    CK(hipDeviceSynchronize());
    double best = 1e30, best_host = 0.0;
    for (int rep = 0; rep < 3; ++rep) {
      const unsigned base = rep * (unsigned)(iters + 8);
      auto t0 = std::chrono::steady_clock::now();
      double host_us = 0.0;
      for (int i = 1; i <= iters; ++i) {
        const unsigned tag = base + (unsigned)i;
        auto h0 = std::chrono::steady_clock::now();
        if (v == 4) {
          CK(hipGraphLaunch(ga_f.e, s));
          bump<<<1, 1, 0, side>>>(side_it);
          poll_flag<<<1, 64, 0, side>>>(flag_a, side_it, timeouts);
          busy<<<64, 256, 0, side>>>(20 * us, sink);
          store_flag<<<1, 64, 0, side>>>(flag_t, side_it);
          CK(hipGraphLaunch(gb.e, s));
          CK(hipGraphLaunch(gk_p.e, s));
        } else {
          CK(hipGraphLaunch(ga.e, s));
          if (v == 1) busy<<<64, 256, 0, s>>>(20 * us, sink);
          if (v == 2 || v == 5) {
            CK(hipEventRecord(ev_a, s));
            CK(hipStreamWaitEvent(side, ev_a, 0));
            busy<<<64, 256, 0, side>>>(20 * us, sink);
            CK(hipEventRecord(ev_t, side));
          }
          if (v == 3) {
            CK(hipStreamWriteValue32(s, sig_a, tag, 0));
            CK(hipStreamWaitValue32(side, sig_a, tag, hipStreamWaitValueGte, 0xffffffffu));
            busy<<<64, 256, 0, side>>>(20 * us, sink);
            CK(hipStreamWriteValue32(side, sig_t, tag, 0));
          }
          if (v == 7) {
            CK(hipStreamWriteValue64(s, sig_a, tag, 0));
            CK(hipStreamWaitValue64(side, sig_a, tag, hipStreamWaitValueGte, ~0ull));
            busy<<<64, 256, 0, side>>>(20 * us, sink);
            CK(hipStreamWriteValue64(side, sig_t, tag, 0));
          }
          CK(hipGraphLaunch(gb.e, s));
          if (v == 7) CK(hipStreamWaitValue64(s, sig_t, tag, hipStreamWaitValueGte, ~0ull));
          if (v == 2) CK(hipStreamWaitEvent(s, ev_t, 0));
          if (v == 3) CK(hipStreamWaitValue32(s, sig_t, tag, hipStreamWaitValueGte, 0xffffffffu));
          CK(hipGraphLaunch(gk.e, s));
        }
        CK(hipEventRecord(lagged[i & 3], s));
        host_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - h0).count();
        if (i > 1) CK(hipEventSynchronize(lagged[(i - 1) & 3]));
      }
      CK(hipStreamSynchronize(s));
      CK(hipStreamSynchronize(side));
      auto t1 = std::chrono::steady_clock::now();
      const double per = std::chrono::duration<double, std::micro>(t1 - t0).count() / iters;
      if (per < best) { best = per; best_host = host_us / iters; }
    }
    unsigned to = 0;
    CK(hipMemcpy(&to, timeouts, 4, hipMemcpyDeviceToHost));
    printf("{\"variant\": \"%s\", \"us_per_iteration\": %.1f, \"host_enqueue_us\": %.1f, \"poll_timeouts\": %u}\n",
           names[v], best, best_host, to);
    fflush(stdout);
  }
  return 0;
}
