// env_switches.h -- debug / test switches from the environment, shared by context.hip and entropy.cc (modular.inc).
#ifndef JXLHIP_ENV_SWITCHES_H_
#define JXLHIP_ENV_SWITCHES_H_
#include <atomic>
#include <mutex>

// ---- debug / test switches from the environment -------------------------------------------------
// Read ONCE per process (first use) into atomics, and again only when a test asks (jxlhip_debug_reload_env): the hot,
// multi-threaded paths -- the Modular channel loops on every runner thread, the group jobs -- never call getenv
// (it races with a host application's setenv, and rounds 3-4 called it several times per DC group and thread).
namespace jxlhip_env {
struct Switches {
  std::atomic<bool> loaded{false};
  std::atomic<bool> wp_general{false};        // JXLHIP_WP_GENERAL: Modular channels through the general loop only
  std::atomic<bool> dc_tree{false};           // JXLHIP_DC_TREE (with -DJXLHIP_DC_TIMING): print the channel's tree
  std::atomic<bool> codestream_verbose{false};
  std::atomic<bool> no_pipeline{false};       // JXLHIP_NO_PIPELINE: DC groups, then AC groups (two runner calls)
  std::atomic<long long> test_range_group{-1};  // JXLHIP_TEST_RANGE_GROUP: fault injector of tests/test_codestream.py
  std::atomic<int> multi_interior_first{1};
  std::mutex mu;
};
extern Switches g;  // defined in entropy.cc
void LoadLocked();  // entropy.cc
static inline const Switches& Get() {
  if (!g.loaded.load(std::memory_order_acquire)) {
    std::lock_guard<std::mutex> lock(g.mu);
    if (!g.loaded.load(std::memory_order_relaxed)) LoadLocked();
  }
  return g;
}
}  // namespace jxlhip_env
#endif  // JXLHIP_ENV_SWITCHES_H_
