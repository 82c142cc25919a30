// Shared by the translation units of libmpmae_hip.so (capi.hip, capi_gemm.hip, capi_rs.hip - compiled in parallel by
// __graft_entry__.build()): the launch-program recorder every kernel launch goes through, the launch-status accumulator, the
// library options and a few host helpers. Kernel headers with non-template kernels are included by exactly ONE unit each.
#pragma once
#include "common.cuh"
#include <hip/hip_ext.h>

#define S_(s) reinterpret_cast<hipStream_t>(s)

// ------------------------------------------------------------------------------------------
// Launch programs. Every kernel launch of this library goes through LAUNCH(): normally it is
// issued at once; while a program is being recorded on this thread (mpmae_program_begin_op) the
// fully-resolved launch (kernel, grid, block, LDS, by-value arguments) is appended to the program
// instead. mpmae_program_run() replays a recorded step from C with one HIP stream per lane and
// event ordering between lanes — no Python, no ctypes marshalling, no graph instantiation.
// ------------------------------------------------------------------------------------------
#include <functional>
#include <vector>
struct ProgOp {
  int lane = 0, signal = 0;
  std::vector<int> waits;
  std::vector<std::function<void(hipStream_t)>> launches;
};
struct MpmaeProgram {
  std::vector<ProgOp> ops;
  std::vector<hipStream_t> side;          // lanes 1..n
  std::vector<hipEvent_t> events;         // by signal id
  std::vector<unsigned> epoch;            // run in which events[id] was last recorded
  std::vector<hipEvent_t> join;
  hipEvent_t fork = nullptr;
  unsigned run = 0;
  int nlanes = 1;
  std::vector<int> sig_op, sig_lane;      // by signal id: index / lane of the op that records it (program_end)
  std::vector<char> waited;               // by signal id: some op of another lane waits for it
  std::vector<hipStream_t> lanes_checked_for;   // main streams the side lanes were probed against (see lanes_overlap_check), at most 8
};
extern thread_local MpmaeProgram* g_rec;      // (defined in capi.hip)

template <typename F>
static inline void submit(hipStream_t st, F&& f) {
  if (g_rec) g_rec->ops.back().launches.emplace_back(std::forward<F>(f));
  else f(st);
}
// Launch status: hipGetLastError() is a per-thread STICKY value that any earlier runtime call of the process may have set (PyTorch
// probes pointers / pinned memory and leaves hipErrorInvalidValue behind: seen as flaky "launch failed" returns of whichever entry
// point ran next). Every launch therefore clears the stale value first and folds ITS OWN status into a thread-local accumulator that
// the entry point returns and resets (launch_status()).
extern thread_local int g_launch_err;
static inline int launch_status() { const int e = g_launch_err; g_launch_err = 0; return e; }
// g_stop_ev: set by mpmae_program_run in front of the LAST launch of an op whose cross-lane signal some other lane waits for
// (MPMAE_OPT_EVX): that launch takes the event as its own completion event (hipExtLaunchKernelGGL stopEvent - the dispatch packet's
// completion signal) instead of a hipEventRecord behind it, which is a barrier packet of its own in the lane's queue: 6.99 -> 5.34 us per
// dependent small kernel on the signalling lane, plain 4.03 (tools/probes/ext_event_probe.hip). A submit that is not a kernel launch leaves
// the event untouched and the replay loop records it the old way.
extern thread_local hipEvent_t g_stop_ev;      // (defined in capi.hip)
#define LAUNCH(kern, g, b, lds, st, ...) \
  submit(st, [=](hipStream_t st__) { (void)hipGetLastError(); \
                                     if (g_stop_ev) { hipEvent_t ev__ = g_stop_ev; g_stop_ev = nullptr; \
                                                      hipExtLaunchKernelGGL(kern, g, b, lds, st__, nullptr, ev__, 0, __VA_ARGS__); } \
                                     else hipLaunchKernelGGL(kern, g, b, lds, st__, __VA_ARGS__); \
                                     const hipError_t e__ = hipGetLastError(); if (e__ != hipSuccess && !g_launch_err) g_launch_err = (int)e__; })
#define RET() return launch_status()

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
static inline int grid1d(long long total, int per_block = 256, int cap = 16384) {
  long long g = (total + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}


// second stage of the two-stage reductions (reduce_partials_kernel in misc.cuh; defined in capi.hip)
void launch_reduce(int mode, const float* part, int P, int W, float* out, float* out2, int a, int b, int c, int d, hipStream_t st);
void launch_reduce_rowscale(const float* part, int P, int W, float* out, float* out2, int nk, int Kk, const float* rs, hipStream_t st);
extern int g_opt[MPMAE_OPT_COUNT_];      // library options (mpmae_set_option; capi.hip)
int ps_num_cus();                        // (capi_rs.hip)
