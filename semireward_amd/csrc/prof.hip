// Kernel-execution timing of single launches for bench.py's roofline object.  While profiling is on, every launch of the library goes through
// hipExtLaunchKernel with a (start, stop) event pair bound to THAT dispatch: the stop event carries the kernel's own end timestamp, the start
// event is a marker directly in front of the dispatch, so the pair spans the kernel's execution plus the 3-4 us between marker and first wave
// (measured against rocprofv3's durations of the same launches in one process: +4 %, profiles/r05_roofline_vs_rocprof.txt).  That is 8 us closer
// to the kernel than an event pair recorded around the host call, needs no calibration, and errs on the conservative side only.  (semilearn has no counterpart: it times whole iterations with two CUDA events and a
// synchronisation, semilearn/core/hooks/timer.py.)
#include <hip/hip_runtime.h>

#include <vector>

#include "common.h"

bool g_sr_prof_on = false;

namespace {
std::vector<hipEvent_t> g_ev;      // 2 per recorded launch: start, stop
int g_n = 0;                       // launches recorded since srhip_prof_enable(1)
}  // namespace

bool sr_prof_take(hipEvent_t* es, hipEvent_t* ee) {
  if ((size_t)(2 * g_n + 2) > g_ev.size()) {
    const size_t want = g_ev.size() + 1024;
    while (g_ev.size() < want) {
      hipEvent_t e;
      if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return false; }
      g_ev.push_back(e);
    }
  }
  *es = g_ev[2 * g_n];
  *ee = g_ev[2 * g_n + 1];
  ++g_n;
  return true;
}

// on != 0: start recording (the launch counter restarts at 0; the events are reused); on == 0: stop.  Returns the number of launches recorded so far.
extern "C" int srhip_prof_enable(int on) {
  const int n = g_n;
  g_sr_prof_on = on != 0;
  if (on) g_n = 0;
  return n;
}

extern "C" int srhip_prof_count(void) { return g_n; }

// Sum of the execution times (ms) of the recorded launches [first, last) into *ms_sum (HOST pointer).  The caller has synchronised the device.
extern "C" int srhip_prof_elapsed_ms(int first, int last, float* ms_sum) {
  if (!ms_sum || first < 0 || last > g_n || first > last) return SR_EINVAL;
  float acc = 0.f;
  for (int i = first; i < last; ++i) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, g_ev[2 * i], g_ev[2 * i + 1]) != hipSuccess) { (void)hipGetLastError(); return SR_EINVAL; }
    acc += ms;
  }
  *ms_sum = acc;
  return SR_OK;
}
