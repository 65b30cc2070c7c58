// TEST INFRASTRUCTURE ONLY: stand-in for <hwy/foreach_target.h> (see base.h).
// Real Highway re-includes HWY_TARGET_INCLUDE once per extra SIMD target; this
// shim has exactly one (static, single-lane) target, so there is nothing to do.
