#!/usr/bin/env python3
"""Where do the cycles of the self-correcting predictor's loop go ON THIS CPU?  Builds patched copies of
libjxl_amd/csrc/modular.inc (timing experiments, most of them decode garbage on purpose), links each into a library of
its own under /tmp and prints the channel times of the first DC groups of tests/data/e2e_8k_d1.jxl.
usage: tools/r04/wp_variants.py [variant ...]"""
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "libjxl_amd", "csrc")
HEALTH = """    if (!br.Healthy()) {
      *br_io = br;
      reader->SetState(state);
      return kBad;
    }
  }
  *br_io = br;
  reader->SetState(state);
  return kOk;
}

// ---- channels whose subtree"""
NO_HEALTH = """  }
  *br_io = br;
  reader->SetState(state);
  return kOk;
}

// ---- channels whose subtree"""
CLONES = '#define JXLHIP_X86_64_V3_CLONE __attribute__((target_clones("default", "arch=x86-64-v3")))'
ANS_A = "      br.Refill();\n      const uint32_t res = state & (kAnsTab - 1);\n      const AliasEntry e = alias[((size_t)ctx << log_alpha) + (res >> log_entry)];"
ANS_B = "      const int64_t N8 = N * 8, W8 = W * 8"


def cut_ans(s):
    a = s.index(ANS_A)
    b = s.index(ANS_B, a)
    return s[:a] + "      uint32_t token = ctx & 1;\n" + s[b:]


VARIANTS = {
    "current": (lambda s: s, []),
    "no_clones": (lambda s: s.replace(CLONES, "#define JXLHIP_X86_64_V3_CLONE"), []),
    "bmi2_clone": (lambda s: s.replace(CLONES, '#define JXLHIP_X86_64_V3_CLONE __attribute__((target_clones("default", "bmi2")))'), []),
    "native": (lambda s: s.replace(CLONES, "#define JXLHIP_X86_64_V3_CLONE"), ["-march=native"]),
    "no_slp": (lambda s: s, ["-fno-slp-vectorize"]),
    "no_slp_no_cmovconv": (lambda s: s, ["-fno-slp-vectorize", "-mllvm", "-x86-cmov-converter=false"]),
    "no_slp_O2": (lambda s: s, ["-fno-slp-vectorize", "-O2"]),
    "no_slp_no_ans": (lambda s: cut_ans(s).replace(HEALTH, NO_HEALTH), ["-fno-slp-vectorize"]),
    "no_ans": (lambda s: cut_ans(s).replace(HEALTH, NO_HEALTH), []),
    "const_weights": (lambda s: s.replace("weight(t[0] + twice * e1_0, 0)", "weight(t[0], 0)").replace("weight(t[1] + twice * e1_1, 1)", "weight(t[1], 1)")
                      .replace("weight(t[2] + twice * e1_2, 2)", "weight(t[2], 2)").replace("weight(t[3] + twice * e1_3, 3)", "weight(t[3], 3)")
                      .replace(HEALTH, NO_HEALTH), []),
    "no_t_rmw": (lambda s: s.replace("      t2[0] += e1_0, t2[1] += e1_1, t2[2] += e1_2, t2[3] += e1_3;", "").replace(HEALTH, NO_HEALTH), []),
    "no_e_store": (lambda s: s.replace("      Ec[4 * x] = e1_0, Ec[4 * x + 1] = e1_1, Ec[4 * x + 2] = e1_2, Ec[4 * x + 3] = e1_3;", "").replace(HEALTH, NO_HEALTH), []),
}


def build(name):
    patch, flags = VARIANTS[name]
    d = "/tmp/wpv_" + name
    shutil.rmtree(d, ignore_errors=True)
    shutil.copytree(CSRC, d, ignore=shutil.ignore_patterns("build"))
    os.makedirs(os.path.join(d, "..", "..", "include"), exist_ok=True)
    src = open(os.path.join(CSRC, "modular.inc")).read()
    out = patch(src)
    if name != "current" and out == src and not flags:
        raise SystemExit("variant %s: nothing to patch" % name)
    open(os.path.join(d, "modular.inc"), "w").write(out)
    text = open(os.path.join(d, "entropy.cc")).read().replace('"../../include/', '"%s/include/' % ROOT)
    open(os.path.join(d, "entropy.cc"), "w").write(text)
    obj = "/tmp/wpv_%s.o" % name
    so = "/tmp/libjxl_wpv_%s.so" % name
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-DJXLHIP_DC_TIMING"] + flags +
                          ["-c", os.path.join(d, "entropy.cc"), "-o", obj], stderr=subprocess.DEVNULL)
    objs = [os.path.join(CSRC, "build", f) for f in sorted(os.listdir(os.path.join(CSRC, "build"))) if f.endswith(".o") and f not in ("entropy.o", "runner.o")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs + [obj])
    return so


for name in (sys.argv[1:] or list(VARIANTS)):
    so = build(name)
    env = dict(os.environ, JXLHIP_LIB=so, JXLHIP_WPV_QUIET="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "r04", "dc_bench.py"), "3", "--dc-only-lenient"], env=env, capture_output=True, text=True)
    ts = [float(m) for m in re.findall(r"channel [012] 256x256: self-correcting predictor's own loop: ([0-9.]+) ms", r.stderr + r.stdout)]
    ts.sort()
    print("%-14s 256x256 channels: best %.2f  median %.2f ms  (%d samples)" % (name, ts[0] if ts else -1, ts[len(ts) // 2] if ts else -1, len(ts)), flush=True)
