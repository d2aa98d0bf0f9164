"""Generates build/ubench_issue.hip: FP64 VALU ISSUE-CADENCE micro-benchmarks for gfx950 with explicit registers.

Question (round 5): the Euler loops run at ~4.45 shader cycles per VALU instruction per SIMD (PMC: 370 cycles for 83 VALU
per wave-step, six waves per SIMD), the compiler-allocated FMA benchmark of tools/ubench_fp64.hip at ~4.8 -- against the 4.0
a 16-lane SIMD needs for a wave64 FP64 operation.  Is the difference a property of WHICH registers an instruction reads
(VGPR bank conflicts between the 64-bit sources), of the number of VGPR sources, of the instruction kind, or of the number of
waves?  Every variant below is one inline-asm block: explicit registers, U independent accumulators, `iters` iterations,
timed per wave with s_memtime (shader cycles) and s_memrealtime (100 MHz).

    python tools/gen_ubench_issue.py && hipcc --offload-arch=gfx950 -O2 build/ubench_issue.hip -o build/ubench_issue
    build/ubench_issue > gpurun_out/ubench_issue.txt          (on the GPU box)
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
U = 8


def vp(r):
    return "v[%d:%d]" % (r, r + 1)


def variants():
    out = []

    def acc(bank, i, stride=4):
        return 8 + bank + stride * i        # v8.. (bank 0) / v10.. (bank 2), stride 4 keeps the bank

    # A: in-place fma acc = acc * B + C, all three sources VGPR pairs, banks (acc, B, C) in {0, 2}
    for p in (0, 2):
        for q in (0, 2):
            for r in (0, 2):
                body = ["v_fma_f64 %s, %s, %s, %s" % (vp(acc(p, i)), vp(acc(p, i)), vp(40 + q), vp(44 + r)) for i in range(U)]
                out.append(("fma acc%d B%d C%d" % (p, q, r), body))
    # A': natural allocation, accumulators at stride 2 (banks alternate)
    body = ["v_fma_f64 %s, %s, %s, %s" % (vp(8 + 2 * i), vp(8 + 2 * i), vp(40), vp(46)) for i in range(U)]
    out.append(("fma acc stride2 B0 C2", body))
    # B: one SGPR source
    for p in (0, 2):
        for r in (0, 2):
            body = ["v_fma_f64 %s, %s, s[60:61], %s" % (vp(acc(p, i)), vp(acc(p, i)), vp(44 + r)) for i in range(U)]
            out.append(("fma acc%d B=sgpr C%d" % (p, r), body))
    # C: two inline constants
    body = ["v_fma_f64 %s, %s, 1.0, 0" % (vp(acc(0, i)), vp(acc(0, i))) for i in range(U)]
    out.append(("fma acc0 B=1.0 C=0 (one VGPR source)", body))
    # C': the accumulator as src2 (a*b + acc): which operand slot matters?
    for q, r in ((0, 0), (0, 2), (2, 0)):
        body = ["v_fma_f64 %s, %s, %s, %s" % (vp(acc(0, i)), vp(40 + q), vp(44 + r), vp(acc(0, i))) for i in range(U)]
        out.append(("fma X%d Y%d acc0(src2)" % (q, r), body))
    # D: mul / add (two sources)
    for op in ("v_mul_f64", "v_add_f64"):
        for p, q in ((0, 0), (0, 2), (2, 0), (2, 2)):
            body = ["%s %s, %s, %s" % (op, vp(acc(p, i)), vp(acc(p, i)), vp(40 + q)) for i in range(U)]
            out.append(("%s acc%d B%d" % (op[2:5], p, q), body))
        body = ["%s %s, %s, 1.0" % (op, vp(acc(0, i)), vp(acc(0, i))) for i in range(U)]
        out.append(("%s acc0 B=1.0" % op[2:5], body))
    # E: no dependence at all, dst apart from the sources
    for q, r, s in ((0, 0, 0), (0, 2, 0), (0, 2, 2), (0, 0, 2)):
        body = ["v_fma_f64 %s, %s, %s, %s" % (vp(acc(0, i)), vp(40 + q), vp(44 + r), vp(80 + s)) for i in range(U)]
        out.append(("fma dst0 = X%d*Y%d+Z%d (no dependence)" % (q, r, s), body))
    # F: same register twice (x*x + c): two distinct reads
    body = ["v_fma_f64 %s, %s, %s, %s" % (vp(acc(0, i)), vp(40), vp(40), vp(44)) for i in range(U)]
    out.append(("fma dst0 = X0*X0+Z0", body))
    # G: alternating kinds, the loop's mix (fma, mul, add)
    body = []
    for i in range(U):
        op = ("v_fma_f64", "v_mul_f64", "v_fma_f64", "v_add_f64")[i % 4]
        if op == "v_fma_f64":
            body.append("v_fma_f64 %s, %s, %s, %s" % (vp(acc(0, i)), vp(acc(0, i)), vp(42), vp(44)))
        else:
            body.append("%s %s, %s, %s" % (op, vp(acc(0, i)), vp(acc(0, i)), vp(42)))
    out.append(("mix fma/mul/fma/add acc0 B2 C0", body))
    # H: 32-bit integer VALU for scale (one dword source each)
    body = ["v_add_u32 v%d, v%d, v%d" % (8 + i, 8 + i, 40) for i in range(U)]
    out.append(("v_add_u32 (32-bit, for scale)", body))
    # I: trans
    body = ["v_rcp_f64 %s, %s" % (vp(acc(0, i)), vp(acc(0, i))) for i in range(U)]
    out.append(("v_rcp_f64 in place", body))
    return out


def kernel(idx, body):
    init = []
    for r in list(range(8, 8 + 4 * U + 4, 2)) + [40, 42, 44, 46, 48, 50]:
        init.append("v_mov_b32 v%d, 0" % r)
        init.append("v_mov_b32 v%d, 0x3ff00000" % (r + 1))
    for r in (44, 46):  # C = 0.0
        init.append("v_mov_b32 v%d, 0" % (r + 1))
    lines = ["s_mov_b32 s50, %4", "s_mov_b32 s60, 0", "s_mov_b32 s61, 0x3ff00000"] + init + [
        "s_nop 4", "s_memtime s[52:53]", "s_memrealtime s[54:55]", "s_waitcnt lgkmcnt(0)", "L_loop_%=:"]
    lines += body * 4  # 4 x unrolled: loop control is < 1 % of the instructions
    lines += ["s_sub_u32 s50, s50, 1", "s_cmp_lg_u32 s50, 0", "s_cbranch_scc1 L_loop_%=", "s_nop 4", "s_memtime s[56:57]", "s_memrealtime s[58:59]",
              "s_waitcnt lgkmcnt(0)", "s_mov_b64 %0, s[52:53]", "s_mov_b64 %1, s[56:57]", "s_mov_b64 %2, s[54:55]", "s_mov_b64 %3, s[58:59]"]
    clob = ['"v%d"' % r for r in range(8, 52)] + ['"s%d"' % r for r in range(50, 62)] + ['"scc"', '"memory"']
    asm = "\n".join('      "%s\\n"' % ln for ln in lines)
    return """__global__ __launch_bounds__(256) void k%d(unsigned long long *out, int iters) {
  unsigned long long t0, t1, r0, r1;
  asm volatile(
%s
      : "=&s"(t0), "=&s"(t1), "=&s"(r0), "=&s"(r1) : "s"(iters) : %s);
  if ((threadIdx.x & 63) == 0) {
    const unsigned w = blockIdx.x * 4 + (threadIdx.x >> 6);
    out[2 * w] = t1 - t0;
    out[2 * w + 1] = r1 - r0;
  }
}
""" % (idx, asm, ", ".join(clob))


def main():
    vs = variants()
    src = ["// generated by tools/gen_ubench_issue.py -- do not edit", "#include <hip/hip_runtime.h>", "#include <algorithm>", "#include <cstdio>",
           "#include <vector>", ""]
    for i, (_, body) in enumerate(vs):
        src.append(kernel(i, body))
    src.append("typedef void (*kern_t)(unsigned long long *, int);")
    src.append("static const struct { const char *name; kern_t k; } K[] = {")
    for i, (name, _) in enumerate(vs):
        src.append('  {"%s", k%d},' % (name, i))
    src.append("};")
    src.append("""
int main(int argc, char **argv) {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) { printf("no device\\n"); return 1; }
  const int cus = p.multiProcessorCount, iters = 16000, per_iter = %d * 4;
  printf("# %%s %%s CUs=%%d; every variant: %%d independent operations per iteration x 4 (unrolled), %%d iterations; W = waves per SIMD\\n", p.name, p.gcnArchName, cus,
         %d, iters);
  printf("# cyc/op/SIMD = HIP-event time of the launch x MHz / (wave-operations per SIMD); wave view = a wave's own s_memtime ticks / its operations / W (median); MHz = s_memtime ticks / s_memrealtime time (median)\\n");
  unsigned long long *d;
  (void)hipMalloc(&d, sizeof(unsigned long long) * 2 * cus * 8 * 4);
  std::vector<unsigned long long> h(2 * cus * 8 * 4);
  const int Ws[] = {8, 4, 2, 1};
  for (unsigned v = 0; v < sizeof(K) / sizeof(K[0]); ++v) {
    printf("%%-44s", K[v].name);
    for (int W : Ws) {
      const int blocks = cus * W, waves = blocks * 4;
      hipLaunchKernelGGL(K[v].k, dim3(blocks), dim3(256), 0, 0, d, 64);
      (void)hipDeviceSynchronize();
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(K[v].k, dim3(blocks), dim3(256), 0, 0, d, iters);
      (void)hipEventRecord(e1);
      if (hipEventSynchronize(e1) != hipSuccess) { printf("launch failed\\n"); return 1; }
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost);
      std::vector<double> cyc(waves), mhz(waves);
      for (int w = 0; w < waves; ++w) { cyc[w] = (double)h[2 * w]; mhz[w] = (double)h[2 * w] / ((double)h[2 * w + 1] / 100.0); }
      std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
      const double ops = (double)iters * per_iter;
      /* two views: (a) the launch as a whole -- kernel time x measured clock / wave-operations per SIMD (includes ~10 us of ramp
       * and tail); (b) a wave's own elapsed cycles / its operations / W (valid when all W waves of a SIMD are resident together) */
      const double cyc_launch = (double)ms * 1e-3 * mhz[waves / 2] * 1e6 / (ops * W);
      printf("  W=%%d: %%5.2f cyc/op/SIMD (wave view %%5.2f) %%4.0f MHz |", W, cyc_launch, cyc[waves / 2] / ops / W, mhz[waves / 2]);
      (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    printf("\\n");
  }
  return 0;
}
""" % (U, U))
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    path = os.path.join(ROOT, "build", "ubench_issue.hip")
    open(path, "w").write("\n".join(src))
    print(path, len(vs), "variants")


if __name__ == "__main__":
    main()
