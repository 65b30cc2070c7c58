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
  std::atomic<bool> codestream_verbose{false};
  std::atomic<bool> no_pipeline{false};       // JXLHIP_NO_PIPELINE: DC groups, then AC groups (two runner calls)
  std::atomic<long long> test_range_group{-1};  // JXLHIP_TEST_RANGE_GROUP: fault injector of tests/test_codestream.py
  std::atomic<int> multi_interior_first{1};
  // launch geometry / path switches of the kernels (experiments and the parity tests; kUnset = the built-in default).
  // Sampled when a context is created (jxlhip_create_ex reloads), never on a launch.
  static constexpr int kUnset = -2147483647 - 1;
  std::atomic<int> fused_pc_rh{0};          // JXLHIP_FUSED_PC_RH: rows per window chunk (0 = fill the device)
  std::atomic<int> filter_rh{0};            // JXLHIP_FILTER_RH: rows per wave of the two-phase filter march (0 = fill the device)
  std::atomic<int> big_wgs{kUnset};         // JXLHIP_BIG_WGS: workgroups of the 64-point family in k_transform_r
  std::atomic<bool> multi_force_gather{false};  // JXLHIP_MULTI_FORCE_GATHER: one-device boxes exercise the gather copies
  std::atomic<unsigned long long> max_pixels{1ull << 30};  // JXLHIP_MAX_PIXELS: what jxlhip_decode_codestream allocates for at most
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
