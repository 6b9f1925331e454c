"""Timing-only knock-out variants of the forward GEMM kernel (wrong numerics on purpose): which resource bounds the loop?
Writes unast_amd/csrc/build_exp/gemm_<v>.hip and builds unast_amd/libunast_hip_<v>.so (both git-ignored)."""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cs = os.path.join(root, "unast_amd", "csrc")
src = open(os.path.join(cs, "gemm.hip")).read()
# only the kernel under study is instantiated (fast builds)
a = src.index("    if (a_mode == OP_KC && b_mode == OP_KC) launch_split")
b = src.index("    if (p.slab) {", a)
only = '''    if (a_mode == OP_KC && b_mode == OP_KC && nsplit == 3 && (flags & 1)) {
        if (flags & 2) hipLaunchKernelGGL((gemm_kernel<OP_KC, OP_KC, 3, 2, 4, 3>), grid, dim3(512), 0, stream, p);
        else hipLaunchKernelGGL((gemm_kernel<OP_KC, OP_KC, 3, 2, 4, 1>), grid, dim3(512), 0, stream, p);
    } else return unast_set_error(UNAST_ERR_ARG, "experiment build: only the interior forward kernel exists");
'''
base = src[:a] + only + src[b:]
a = base.index("template <int AM, int BMODE, int FLAGS>\nstatic void launch_tile")
b = base.index("static bool aligned16")
base = base[:a] + base[b:]
V = {"base": lambda s: s}
def no_lds_write(s):
    s = s.replace("*reinterpret_cast<u32x2*>(sA + off) = hi;", "if (p.K < 0) *reinterpret_cast<u32x2*>(sA + off) = hi;")
    s = s.replace("if (PARTS == 2) *reinterpret_cast<u32x2*>(sA + A_BYTES + off) = lo;", "if (PARTS == 2 && p.K < 0) *reinterpret_cast<u32x2*>(sA + A_BYTES + off) = lo;")
    s = s.replace("*reinterpret_cast<u32x2*>(sB + off) = hi;", "if (p.K < 0) *reinterpret_cast<u32x2*>(sB + off) = hi;")
    s = s.replace("if (PARTS == 2) *reinterpret_cast<u32x2*>(sB + B_BYTES + off) = lo;", "if (PARTS == 2 && p.K < 0) *reinterpret_cast<u32x2*>(sB + B_BYTES + off) = lo;")
    return s
def few_mfma(s):
    return s.replace("            for (int j = 0; j < 4; ++j) {\n                // swapped operands", "            for (int j = 0; j < 4; ++j) {\n                if (j > 0 && p.K > 0) continue;\n                // swapped operands")
def no_gload(s):
    return s.replace("        if (FULL) {     // prefetches past", "        if (kt > 3 && p.K > 0) return;\n        if (FULL) {     // prefetches past")
def no_lds_read(s):
    s = s.replace("    auto compute = [&](const unsigned char* sA, const unsigned char* sB) {\n        bf16x8_t af[MI][PARTS], bfr[4][PARTS];",
                  "    bf16x8_t af[MI][PARTS], bfr[4][PARTS];\n    bool first = true;\n    auto compute = [&](const unsigned char* sA, const unsigned char* sB) {\n        if (first || p.K < 0) {")
    s = s.replace("            for (int i = 0; i < 4; ++i) bfr[i][s] = frag(sB + s * B_BYTES, B_KC, B_RCS, wn * 64 + i * 16);\n        }\n",
                  "            for (int i = 0; i < 4; ++i) bfr[i][s] = frag(sB + s * B_BYTES, B_KC, B_RCS, wn * 64 + i * 16);\n        }\n        first = false; }\n")
    return s
def no_barrier(s):
    a = s.index("    for (int kt = 0; kt < nk; kt += 2) {\n        load_tiles(kt + 2, ra0, rb0);\n        compute(s0A, s0B);")
    b = s.index("    if (AM == OP_RC && do_rowsum) {", a)
    return s[:a] + s[a:b].replace("__syncthreads();", "if (p.K < 0) __syncthreads();") + s[b:]
def no_use(s):      # loads are issued but their data is never waited for (tile content = constants)
    return s.replace("            split4<NSPLIT>(ra[i], hi, lo);\n            const int off = A_KC ?", "            float4 cst = make_float4(1.f, 2.f, 3.f, (float)i); split4<NSPLIT>(p.K < 0 ? ra[i] : cst, hi, lo);\n            const int off = A_KC ?").replace(
        "            if (BSPLIT) {\n                hi[0] = __float_as_uint(rb[i].x);", "            if (p.K > 0) { hi[0] = 1u; hi[1] = 2u; lo[0] = 3u; lo[1] = i; } else if (BSPLIT) {\n                hi[0] = __float_as_uint(rb[i].x);", 1)
def no_store(s):
    return s.replace("                *reinterpret_cast<float4*>(cp) = o;", "                if (p.K < 0) *reinterpret_cast<float4*>(cp) = o;")
V.update(nostore=no_store, nobarrier=no_barrier, nouse=no_use, nouse_nobarrier=lambda s: no_barrier(no_use(s)),
         skeleton=lambda s: no_barrier(no_use(no_lds_read(no_lds_write(few_mfma(s))))),
         nowrite=no_lds_write, fewmfma=few_mfma, nogload=no_gload, noread=no_lds_read,
         nowrite_noread=lambda s: no_lds_read(no_lds_write(s)), nogload_nowrite=lambda s: no_gload(no_lds_write(s)))
which = sys.argv[1:] or list(V)
objs = [os.path.join(cs, "build", n + ".o") for n in ("attention", "elementwise", "loss", "lstm", "norm", "optim", "api")]
for v in which:
    s = V[v](base)
    assert v == "base" or s != base, v
    f = os.path.join(cs, "build_exp", "gemm_%s.hip" % v)
    open(f, "w").write(s.replace('#include "common.h"', '#include "../common.h"'))
    o = f[:-4] + ".o"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", "-c", f, "-o", o])
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950"] + objs + [o, "-o", os.path.join(root, "unast_amd", "libunast_hip_%s.so" % v)])
    print("built", v, flush=True)
