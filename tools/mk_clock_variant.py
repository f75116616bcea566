"""dev: build tools/_variants/clock.so — head/tail kernels with wall_clock64 stamps at phase boundaries (read by tools/phase_clock.py)."""
import os, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(R + '/psi-release_amd/csrc/fit.hip').read()
blk = '''#ifdef PSI_PHASE_CLOCK
__device__ unsigned long long g_dbg[64];
#define PHASE(i) do { __syncthreads(); if (b == 0 && last && threadIdx.x == 0) g_dbg[i] = wall_clock64(); } while (0)
#define PHASEB(i) do { __syncthreads(); if (b == 0 && threadIdx.x == 0) atomicMax(&g_dbg[i], (unsigned long long)wall_clock64()); } while (0)
#else
#define PHASE(i)
#endif
'''
def after(marker, n, s):
    assert marker in s, marker
    return s.replace(marker, marker + "\n    PHASE(%d);" % n, 1)
s = s.replace('#include "psi_internal.h"\n#include "lbs_device.h"\n', '#include "psi_internal.h"\n' + blk + '#include "/tmp/clk_inc/lbs_device.h"\n', 1)
s = s.replace("    const bool last = c == C - 1;           // the workgroup that carries on after the exchange", "    const bool last = c == C - 1;\n    PHASE(0);")
s = after("        psi_pose_fwd_rest(lv.m, sbetas, sJ);\n    }", 1, s)
s = after("        for (int e = 0; e < 4; e++) sh1[t * 4 + e] = leaky(a[e], 0.2f);\n    }\n    __syncthreads();", 2, s)
s = after("        for (int o = t; o < NH; o += HB) f.h1[(size_t)b * NH + o] = sh1[o];\n    __syncthreads();", 3, s)
s = after("        part3[ks3][og3] = a;\n    }\n    __syncthreads();", 4, s)
s = after("    if (C > 1 && !last) return;\n    __syncthreads();", 8, s)
s = after("        for (int e = 0; e < 3; e++) spose[t * 3 + e] = aa[e] + pm[e];\n    }\n    __syncthreads();", 5, s)
s = after("    psi_pose_fwd_chain(lv.m, spose, nullptr, f.B, b, sJ, par, lvl, lv.feat, lv.R, lv.G, lv.A, nullptr);", 6, s)
s = s.replace("    psi_pose_bwd_body(lv.m, f.pose + (size_t)b * f.J * 3, lv.R,", "    PHASE(16);\n    psi_pose_bwd_body(lv.m, f.pose + (size_t)b * f.J * 3, lv.R,")
s = after("    __syncthreads();                                         // g_betas / g_pose / g_rot of this body are in LDS", 18, s)
s = after("        sgx[9 + (t - 160)] = sgbetas[t - 160];\n    }\n    __syncthreads();", 19, s)
s = after("        sga2[t] = a * (h2v > 0.0f ? 1.0f : 0.2f);\n    }\n    __syncthreads();", 20, s)
s = after("        part4[kq2 * 128 + og2] = a0 + a1;\n    }\n    __syncthreads();", 21, s)
s = after("            sga1[t] = (o + a) * (h1v > 0.0f ? 1.0f : 0.2f);\n        }\n    }\n    __syncthreads();", 25, s)
s = after("        for (int e = 0; e < 4; e++) sgx[19 + t * 4 + e] = a[e];\n    }\n    __syncthreads();", 22, s)
s = s.replace("        f.x[o] = sx[t] - step_size * (m / denom);\n    }\n}", "        f.x[o] = sx[t] - step_size * (m / denom);\n    }\n    PHASE(23);\n}")
s += """
#ifdef PSI_PHASE_CLOCK
extern "C" int psi_dbg_read(unsigned long long *h) { return (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 64); }
#endif
"""
open('/tmp/fit_clock.hip', 'w').write(s)
d = open(R + '/psi-release_amd/csrc/lbs_device.h').read()
d = d.replace("#pragma once", "#pragma once\n#ifndef PHASEB\n#define PHASEB(i)\n#endif", 1)
def rep(a, b):
    global d
    assert a in d, a
    d = d.replace(a, b, 1)
rep("    __syncthreads();\n    if (act) {\n        for (int c = 0; c < 3; c++) sRel[j][c]",
    "    PHASEB(32);\n    if (act) {\n        for (int c = 0; c < 3; c++) sRel[j][c]")
rep("    // local gradients: gR_j = P_R^T gG_j.R", "    PHASEB(33);\n    // local gradients: gR_j = P_R^T gG_j.R")
rep("    // feature gradient (reduced over n-slices)", "    PHASEB(34);\n    // feature gradient (reduced over n-slices)")
rep("    if (act && (g_pose_b || g_rot_b)) {", "    PHASEB(35);\n    if (act && (g_pose_b || g_rot_b)) {")
# forward pose stage
rep("    __syncthreads();\n    if (act)\n        for (int c = 0; c < 3; c++) Jl[c] = sJ[j][c];",
    "    PHASEB(40);\n    if (act)\n        for (int c = 0; c < 3; c++) Jl[c] = sJ[j][c];")
rep("    if (act) {\n        psi_f4 *Go = (psi_f4 *)(Gs +", "    PHASEB(41);\n    if (act) {\n        psi_f4 *Go = (psi_f4 *)(Gs +")
# skinning forward + SDF epilogue (block (0,0) only)
d = d.replace("#define PHASEB(i)\n#endif", "#define PHASEB(i)\n#endif\n#ifdef PSI_PHASE_CLOCK\n#define PHASE2(i) do { __syncthreads(); if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_dbg[i] = wall_clock64(); } while (0)\n#else\n#define PHASE2(i)\n#endif", 1)
os.makedirs('/tmp/clk_inc', exist_ok=True)
open('/tmp/clk_inc/lbs_device.h', 'w').write(d)
print(subprocess.run([R + '/tools/mkvariant.sh', 'clock', '/tmp/fit_clock.hip', 'fit.hip', '-DPSI_PHASE_CLOCK'], capture_output=True, text=True).stdout[-200:])
