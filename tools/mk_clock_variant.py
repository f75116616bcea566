"""dev: build tools/_variants/clock.so — head/tail kernels with wall_clock64 stamps at phase boundaries (read by tools/phase_clock.py)."""
import os, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = open(R + '/psi-release_amd/csrc/fit.hip').read()
blk = '''#ifdef PSI_PHASE_CLOCK
__device__ unsigned long long g_dbg[64];
#define PHASE(i) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) g_dbg[i] = wall_clock64(); } while (0)
#else
#define PHASE(i)
#endif
'''
def after(marker, n, s):
    assert marker in s, marker
    return s.replace(marker, marker + "\n    PHASE(%d);" % n, 1)
s = s.replace('#include "psi_internal.h"\n#include "lbs_device.h"\n', '#include "psi_internal.h"\n' + blk + '#include "/tmp/clk_inc/lbs_device.h"\n', 1)
s = s.replace("    const int b = blockIdx.x, t = threadIdx.x;\n    __shared__ float sx[XD + 5], sh1[NH], sh2[NH], so6[128], red[HB / 64];",
              "    const int b = blockIdx.x, t = threadIdx.x;\n    PHASE(0);\n    __shared__ float sx[XD + 5], sh1[NH], sh2[NH], so6[128], red[HB / 64];")
s = after("        f.recpart[b] = sr;\n        f.vppart[b] = sz;\n    }", 1, s)
s = after("        for (int c = 0; c < 4; c++) sh1[t * 4 + c] = leaky(a[c], 0.2f);\n    }\n    __syncthreads();", 2, s)
s = after("        for (int c = 0; c < 4; c++) sh2[t * 4 + c] = leaky(a[c], 0.2f);\n    }\n    __syncthreads();", 3, s)
s = after("        for (int c = 0; c < 4; c++) so6[t * 4 + c] = a[c];\n    }\n    __syncthreads();", 4, s)
s = after("    if (t < 3) f.transl[(size_t)b * 3 + t] = sx[t];\n    __syncthreads();                                         // pose / betas20 of this body are visible to the workgroup", 5, s)
s = after("    psi_pose_fwd_body(lv.m, f.betas20, f.pose, f.transl, f.B, b, lv.feat, lv.R, lv.Jl, lv.G, lv.A, nullptr);", 6, s)
s = s.replace("    psi_pose_bwd_body(lv.m, f.betas20, f.pose, lv.R, lv.Jl, lv.G, lv.gA +", "    PHASE(16);\n    PHASE(17);\n    psi_pose_bwd_body(lv.m, f.betas20, f.pose, lv.R, lv.Jl, lv.G, lv.gA +")
s = after("                      f.g_betas, f.g_pose, f.g_rot);", 18, s)
s = after("        sgx[9 + (t - 160)] = f.g_betas[(size_t)b * f.NB + (t - 160)];\n    }\n    __syncthreads();", 19, s)
s = after("        for (int c = 0; c < 4; c++) sga2[t * 4 + c] = a[c] * (h2[t * 4 + c] > 0.0f ? 1.0f : 0.2f);\n    }\n    __syncthreads();", 20, s)
s = after("        for (int c = 0; c < 4; c++) sga1[t * 4 + c] = a[c] * (h1[t * 4 + c] > 0.0f ? 1.0f : 0.2f);\n    }\n    __syncthreads();", 21, s)
s = after("        for (int c = 0; c < 4; c++) sgx[19 + t * 4 + c] = a[c];\n    }\n    __syncthreads();", 22, s)
s = s.replace("        f.x[o] = sx[t] - step_size * (m / denom);\n    }\n}", "        f.x[o] = sx[t] - step_size * (m / denom);\n    }\n    PHASE(23);\n}")
s += '''
#ifdef PSI_PHASE_CLOCK
extern "C" int psi_dbg_read(unsigned long long *h) { return (int)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dbg), sizeof(unsigned long long) * 64); }
#endif
'''
open('/tmp/fit_clock.hip', 'w').write(s)
d = open(R + '/psi-release_amd/csrc/lbs_device.h').read()
d = d.replace("#pragma once", "#pragma once\n#ifndef PHASE\n#define PHASE(i)\n#endif", 1)
def rep(a, b):
    global d
    assert a in d, a
    d = d.replace(a, b, 1)
rep("    __syncthreads();\n    const int par = act ? m.parents[j] : -1;\n    const int lvl = act ? m.level[j] : -1;\n    if (act) {\n        for (int c = 0; c < 3; c++) sRel[j][c]",
    "    PHASE(32);\n    const int par = act ? m.parents[j] : -1;\n    const int lvl = act ? m.level[j] : -1;\n    if (act) {\n        for (int c = 0; c < 3; c++) sRel[j][c]")
rep("    // local gradients: gR_j = P_R^T gG_j.R", "    PHASE(33);\n    // local gradients: gR_j = P_R^T gG_j.R")
rep("    // feature gradient (reduced over n-slices)", "    PHASE(34);\n    // feature gradient (reduced over n-slices)")
rep("    if (act && (g_pose || g_rot)) {", "    PHASE(35);\n    if (act && (g_pose || g_rot)) {")
# forward pose stage
rep("    __syncthreads();\n    const int par = act ? m.parents[j] : -1;\n    const int lvl = act ? m.level[j] : -1;\n    if (act)\n        for (int c = 0; c < 3; c++) Jl[c] = sJ[j][c];",
    "    PHASE(40);\n    const int par = act ? m.parents[j] : -1;\n    const int lvl = act ? m.level[j] : -1;\n    if (act)\n        for (int c = 0; c < 3; c++) Jl[c] = sJ[j][c];")
rep("    if (act) {\n        psi_f4 *Go = (psi_f4 *)(Gs +", "    PHASE(41);\n    if (act) {\n        psi_f4 *Go = (psi_f4 *)(Gs +")
# skinning forward + SDF epilogue (block (0,0) only)
d = d.replace("#define PHASE(i)\n#endif", "#define PHASE(i)\n#endif\n#ifdef PSI_PHASE_CLOCK\n#define PHASE2(i) do { __syncthreads(); if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_dbg[i] = wall_clock64(); } while (0)\n#else\n#define PHASE2(i)\n#endif", 1)
rep("    const int b = blockIdx.y;\n    psi_f2 T2[6];\n    psi_blend_transforms(m, As, b, v, T2);\n    const bool live = v < m.V;", "    const int b = blockIdx.y;\n    PHASE2(48);\n    psi_f2 T2[6];\n    psi_blend_transforms(m, As, b, v, T2);\n    PHASE2(49);\n    const bool live = v < m.V;")
rep("    epi.vertex(b, v, x, y, z, live);\n    epi.finish(b);", "    PHASE2(50);\n    epi.vertex(b, v, x, y, z, live);\n    PHASE2(51);\n    epi.finish(b);\n    PHASE2(52);")
os.makedirs('/tmp/clk_inc', exist_ok=True)
open('/tmp/clk_inc/lbs_device.h', 'w').write(d)
print(subprocess.run([R + '/tools/mkvariant.sh', 'clock', '/tmp/fit_clock.hip', 'fit.hip', '-DPSI_PHASE_CLOCK'], capture_output=True, text=True).stdout[-200:])
