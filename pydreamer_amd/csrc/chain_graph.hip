// Linear hipGraph replay of the library's launch chains (the posterior T loop, the BPTT loop, the imagination H loop).
//
// A chain entry point (rssm.hip) is hundreds of small dependent kernels on ONE stream with arguments that are a pure
// function of the call's arguments.  Enqueued one by one the HIP runtime spends ~6-10 us of host time per launch, which
// is the floor of every small-batch step (data-parallel shards, bf16 operands).  Here the launch sequence of such a call
// is stream-captured once per distinct argument set, instantiated, and replayed with ONE hipGraphLaunch afterwards:
//
//   DmChainGraph cg("rssm_fwd", key, stream);
//   if (cg.replay_only()) return cg.finish();
//   stream = cg.launch_stream();          // the caller's stream, or the recording stream while capturing
//   ... the ordinary launch sequence on `stream` ...
//   return cg.finish();
//
// The key holds every value a kernel argument is computed from (all pointers, the shape, the precision and the step
// range); the torch caching allocator hands a steady-state training loop the same addresses step after step, so the
// cache settles on a handful of entries (pydreamer_amd/models.py keeps the chains' buffers in a per-model arena for this).  A
// cache that keeps missing (addresses that never repeat) switches itself off for that chain - capture + instantiate costs
// more than the eager launches it would save.
//
// What is captured is exactly the eager launch sequence (same kernels, same order, same stream), so results are
// bit-identical; the graph is linear (no branches).  Never under the per-launch profiler (events would be captured as nodes)
// or inside an outer capture (graph.py).
//
// Measured on MI355X / ROCm 7.2 (scripts/chain_graph_bench.py, profiles/r03_chain_graph.txt), idle GPU, one stream:
//   posterior chain B=50 (255 launches): host 2.28 ms eager -> 0.08 ms replay, GPU 3.81 ms either way (15 us per dependent
//   launch); B=7: host 1.21 -> 0.08 ms, GPU 2.00 ms either way; rollout B=50: host 0.95 -> 0.05 ms, GPU 7.0 ms either way.
//   A replayed node costs the GPU what an eager launch costs (the dependent-kernel boundary is the stream's); the host cost
//   drops from ~9 us per launch to ~0.3 us per node.
// In the training step the chains are GPU-latency-bound, not host-bound (the step: 38.5 ms either way, the 7-column shard
// 11.7 ms either way, host enqueue 10.4 -> 9.7 ms), and graph launches issued CONCURRENTLY on two streams from two host threads
// (the pre-launched BPTT chain beside the rollout chain) collapse: 7-column shard 11.7 -> 28.8 ms, full batch 38.5 -> 54.7 ms.
// So the replay is OFF by default; DM_CHAIN_GRAPH=1 / dm_chain_graph_enable(1) switches it on (host-bound deployments:
// small models, slow hosts), DM_CHAIN_GRAPH_ONLY=<names> restricts it to some chains.
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <mutex>
#include <vector>

static std::mutex g_mu;
struct ChainEntry {
  const char* tag;
  DmChainKey key;
  hipGraphExec_t exec;
  uint64_t last_use;
};
struct ChainTagState {
  const char* tag;
  uint64_t hits, misses;
  bool off;
};
static std::vector<ChainEntry> g_cache;
static std::vector<ChainTagState> g_tags;
static uint64_t g_clock = 0;
static const size_t CHAIN_CACHE_CAP = 48;

static int g_chain_graph_on = getenv("DM_CHAIN_GRAPH") ? atoi(getenv("DM_CHAIN_GRAPH")) : 0;
static int chain_graph_enabled() { return g_chain_graph_on; }
// on >= 0 sets the switch (tests / A-B runs: the eager launch sequence and the replay must agree bit for bit); returns the
// previous value.  on < 0 queries.
extern "C" int dm_chain_graph_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  const int prev = g_chain_graph_on;
  if (on >= 0) g_chain_graph_on = on ? 1 : 0;
  return prev;
}
static ChainTagState& tag_state(const char* tag) {      // g_mu held
  for (auto& t : g_tags)
    if (t.tag == tag || strcmp(t.tag, tag) == 0) return t;
  g_tags.push_back(ChainTagState{tag, 0, 0, false});
  return g_tags.back();
}

DmChainGraph::DmChainGraph(const char* tag, const DmChainKey& key, hipStream_t st) : tag_(tag), key_(key), st_(st) {
  if (!chain_graph_enabled() || dm_prof_active() || key.overflow) return;
  static const char* only = getenv("DM_CHAIN_GRAPH_ONLY");      // A/B: comma-separated chain names that may be replayed
  if (only && !strstr(only, tag)) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return;                                   // inside somebody else's capture: our launches simply become its nodes
  }
  {
    std::lock_guard<std::mutex> lk(g_mu);
    ChainTagState& ts = tag_state(tag);
    if (ts.off) return;
    for (auto& e : g_cache)
      if (e.tag == tag && e.key.n == key.n && memcmp(e.key.w, key.w, sizeof(uint64_t) * key.n) == 0) {
        e.last_use = ++g_clock;
        exec_ = e.exec;
        ++ts.hits;
        mode_ = 1;
        return;
      }
    ++ts.misses;
    static const int debug = getenv("DM_CHAIN_GRAPH_DEBUG") ? 1 : 0;
    if (debug) {      // which key words moved since the newest entry of this chain
      const ChainEntry* last = nullptr;
      for (auto& e : g_cache)
        if (e.tag == tag && (!last || e.last_use > last->last_use)) last = &e;
      fprintf(stderr, "[chain graph] %s: miss #%llu (hits %llu)", tag, (unsigned long long)ts.misses, (unsigned long long)ts.hits);
      if (last && last->key.n == key.n) {
        fprintf(stderr, ", words that differ from the newest entry:");
        for (int i = 0; i < key.n; ++i)
          if (last->key.w[i] != key.w[i]) fprintf(stderr, " %d", i);
      }
      fprintf(stderr, "\n");
    }
    // a chain whose arguments never repeat: stop trying (capture + instantiate is dearer than the launches it replaces)
    if (ts.misses >= 24 && ts.hits < ts.misses) { ts.off = true; return; }
  }
  // The launch sequence is recorded on a private stream of this host thread, not on the caller's: torch's default stream is
  // the legacy null stream, which cannot be captured; the instantiated graph is then launched on the caller's stream.
  static thread_local hipStream_t cap = nullptr;
  if (!cap && hipStreamCreateWithFlags(&cap, hipStreamNonBlocking) != hipSuccess) {
    (void)hipGetLastError();
    cap = nullptr;
    return;                                   // eager
  }
  if (hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) != hipSuccess) {
    (void)hipGetLastError();
    return;                                   // eager
  }
  cap_ = cap;
  mode_ = 2;
}

DmChainGraph::~DmChainGraph() {
  if (mode_ == 2) {                           // the body bailed out mid-capture: close it and drop what was recorded
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture(cap_, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
  }
}

int DmChainGraph::finish() {
  if (mode_ == 0) return DM_OK;
  if (mode_ == 1) {
    mode_ = 0;
    // launched under the cache lock: another host thread's finish() may evict (and destroy) this exec otherwise
    std::lock_guard<std::mutex> lk(g_mu);
    bool alive = false;
    for (auto& e : g_cache) alive = alive || e.exec == exec_;
    if (!alive) return dm_fail(DM_E_HIP, "chain graph %s: the cached graph was evicted between lookup and launch", tag_);
    if (hipGraphLaunch(exec_, st_) != hipSuccess) return dm_fail(DM_E_HIP, "chain graph %s: hipGraphLaunch: %s", tag_, hipGetErrorString(hipGetLastError()));
    return DM_OK;
  }
  mode_ = 0;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(cap_, &g);
  if (e != hipSuccess || !g) return dm_fail(DM_E_HIP, "chain graph %s: hipStreamEndCapture: %s", tag_, hipGetErrorString(e));
  hipGraphExec_t ex = nullptr;
  e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess || !ex) return dm_fail(DM_E_HIP, "chain graph %s: hipGraphInstantiate: %s", tag_, hipGetErrorString(e));
  {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_cache.size() >= CHAIN_CACHE_CAP) {             // evict the least recently used entry
      size_t lru = 0;
      for (size_t i = 1; i < g_cache.size(); ++i)
        if (g_cache[i].last_use < g_cache[lru].last_use) lru = i;
      (void)hipGraphExecDestroy(g_cache[lru].exec);
      g_cache.erase(g_cache.begin() + lru);
    }
    g_cache.push_back(ChainEntry{tag_, key_, ex, ++g_clock});
  }
  if (hipGraphLaunch(ex, st_) != hipSuccess) return dm_fail(DM_E_HIP, "chain graph %s: first hipGraphLaunch: %s", tag_, hipGetErrorString(hipGetLastError()));
  return DM_OK;
}

// out[0..3*max_tags): {hits, misses, switched off} per chain, in first-use order; returns the number of chains.
extern "C" int dm_chain_graph_stats(long long* out, int max_tags) {
  std::lock_guard<std::mutex> lk(g_mu);
  int n = 0;
  for (auto& t : g_tags) {
    if (n >= max_tags) break;
    out[3 * n] = (long long)t.hits; out[3 * n + 1] = (long long)t.misses; out[3 * n + 2] = t.off ? 1 : 0;
    ++n;
  }
  return n;
}
// Drops every cached graph (buffers about to be freed, tests); the chains re-capture on their next call.
extern "C" int dm_chain_graph_reset(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& e : g_cache) (void)hipGraphExecDestroy(e.exec);
  g_cache.clear();
  g_tags.clear();
  return DM_OK;
}
