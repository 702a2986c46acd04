r"""Micro-benchmark: what issues BESIDE `v_mfma_f32_32x32x2_f32` on one SIMD of gfx950, and at what price.

Every variant is a generated kernel whose loop body is ONE `asm volatile` block, so the instruction order is exactly the
one written here (round 2's version let the compiler schedule a dependent accumulator chain against a saturating VALU
wave and concluded "VALU never overlaps MFMA").  Measured per variant, on all 256 CUs at once:
`s_memtime` cycles per MFMA of the measured wave (min / median over the waves of the chip) and the wall time.

  python tools/mfma_overlap.py build      # here (hipcc cross-compiles):  tools/_build/mfma_overlap
  python tools/mfma_overlap.py run        # on the GPU box: prints the table (commit it under profiles/)

Groups
  A  one wave per SIMD, MFMA only: 8 accumulators round-robin / chains of 4 per accumulator (the shipped Winograd
     order) / one accumulator
  B  one wave per SIMD, n fillers in every MFMA gap, n = 1 2 4 6 8 12: v_add_f32, v_fma_f32, v_pk_add_f32,
     v_pk_fma_f32, v_mov_b32, s_nop, ds_read_b128, ds_write_b64, buffer_load_dwordx2 (L2-resident) -- for both orders
  C  two waves per SIMD: wave w = MFMA only (round-robin), partner w+4 = fillers only at several densities
     (the partner's achieved rate is reported next to the MFMA wave's cycles per MFMA)
  D  two waves per SIMD, BOTH doing MFMA + n fillers per gap (the Winograd kernel's regime): SIMD cycles per MFMA
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
U = 16  # MFMAs per asm block

# operand numbering of the asm block
#  %0..%7   accumulators c0..c7 (f32x16, "+v")
#  %8..%15  scalar filler registers f0..f7 (float, "+v")
#  %16..%19 packed filler registers p0..p3 (f32x2, "+v")
#  %20..%23 ds_read / buffer_load destinations q0..q3 (float4, "=&v")   (buffer loads write the low half)
#  %24 a, %25 b (float, "v"), %26 lds address (int, "v"), %27 buffer voffset (int, "v"), %28 rsrc ("s")
ACC = ["%%%d" % i for i in range(8)]


def filler(kind: str, i: int) -> str:
    f = "%%%d" % (8 + i % 8)
    p = "%%%d" % (16 + i % 4)
    q = "%%%d" % (20 + i % 4)
    return {
        "v_add": f"v_add_f32 {f}, {f}, %24",
        "v_fma": f"v_fma_f32 {f}, {f}, %25, %24",
        "v_pk_add": f"v_pk_add_f32 {p}, {p}, {p}",
        "v_pk_fma": f"v_pk_fma_f32 {p}, {p}, {p}, {p}",
        "v_mov": f"v_mov_b32 {f}, %24",
        "s_nop": "s_nop 0",
        "ds_read_b128": f"ds_read_b128 {q}, %26 offset:{(i % 8) * 1024}",
        "ds_write_b64": f"ds_write_b64 %26, {p} offset:{(i % 8) * 1024}",
        "buf_load_x2": f"buffer_load_dwordx2 {p}, %27, %28, 0 offen offset:{(i % 8) * 512}",
    }[kind]


def body(order: str, kind: str | None, n: int, mfma: bool = True) -> str:
    lines = []
    k = 0
    for u in range(U):
        if mfma:
            if order == "rr":
                acc = ACC[u % 8]
            elif order == "chain4":
                acc = ACC[(u // 4) % 8]
            elif order == "chain4x8":  # 8 accumulators over two blocks: handled by alternating blocks (see gen)
                acc = ACC[(u // 4) % 8]
            else:
                acc = ACC[0]
            lines.append(f"v_mfma_f32_32x32x2_f32 {acc}, %24, %25, {acc}")
        if kind:
            for _ in range(n):
                lines.append(filler(kind, k))
                k += 1
    if kind in ("ds_read_b128", "ds_write_b64"):
        lines.append(f"s_waitcnt lgkmcnt({min(15, 3 * n)})")
    if kind == "buf_load_x2":
        lines.append(f"s_waitcnt vmcnt({min(15, 3 * n)})")
    return "\\n\\t".join(lines)


OPERANDS = (': "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7), '
            '"+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), '
            '"+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "=&v"(q0), "=&v"(q1), "=&v"(q2), "=&v"(q3) '
            ': "v"(a), "v"(b), "v"(ldsa), "v"(voff), "s"(rsrc) : "memory"')

PRE = r"""
  extern __shared__ float sm[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  f32x16 c0, c1, c2, c3, c4, c5, c6, c7;
  for (int r = 0; r < 16; ++r) { c0[r] = c1[r] = c2[r] = c3[r] = c4[r] = c5[r] = c6[r] = c7[r] = 0.f; }
  float f0 = lane, f1 = lane + 1, f2 = lane + 2, f3 = lane + 3, f4 = lane + 4, f5 = lane + 5, f6 = lane + 6, f7 = lane + 7;
  f32x2 p0 = {f0, f1}, p1 = {f2, f3}, p2 = {f4, f5}, p3 = {f6, f7};
  float4 q0, q1, q2, q3;
  q0 = q1 = q2 = q3 = make_float4(0.f, 0.f, 0.f, 0.f);
  float a = 1.0f + lane * 1e-3f, b = 1.0001f;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) sm[i] = (float)i;
  volatile int* flag = reinterpret_cast<volatile int*>(sm + 16384);
  if (threadIdx.x == 0) *flag = 0;
  __syncthreads();
  int ldsa = (wave * 64 + lane) * 16;                    // 16-byte slots, conflict-free
  int voff = (int)((blockIdx.x * blockDim.x + threadIdx.x) & 1023) * 8;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)gbuf, 0, 1 << 20, 0x00020000);
"""

POST = r"""
  float s = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0.x + p1.y + p2.x + p3.y + q0.x + q1.y + q2.z + q3.w;
  for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r] + c4[r] + c5[r] + c6[r] + c7[r];
  if (s == 1.2345e-30f) out[threadIdx.x] = s;
"""


def kernel(name: str, nthreads: int, body_a: str, body_b: str | None, b_until_flag: bool) -> str:
    r"""waves 0..3 run body_a `iters` times and stamp cycles; waves 4..7 (if any) run body_b until waves 0..3 are done
    (b_until_flag) or `iters` times."""
    src = f"__global__ __launch_bounds__({nthreads}) void {name}(float* out, int iters, unsigned long long* cyc, unsigned* cnt, const float* gbuf) {{\n{PRE}"
    loop_a = (f"    const unsigned long long t0 = __builtin_readcyclecounter();\n"
              f"    for (int it = 0; it < iters; ++it) asm volatile(\"{body_a}\" {OPERANDS});\n"
              f"    asm volatile(\"s_nop 0\" : \"+v\"(c0), \"+v\"(c7));\n"
              f"    const unsigned long long t1 = __builtin_readcyclecounter();\n"
              f"    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;\n")
    if body_b is None:
        src += "  {\n" + loop_a + "  }\n"
    else:
        src += "  if (wave < 4) {\n" + loop_a
        if b_until_flag:
            src += "    if (lane == 0) atomicAdd((int*)flag, 1);\n  } else {\n"
            src += (f"    unsigned n = 0;\n    const unsigned long long t0 = __builtin_readcyclecounter();\n"
                    f"    while (*flag < 4) {{ asm volatile(\"{body_b}\" {OPERANDS}); ++n; }}\n"
                    f"    const unsigned long long t1 = __builtin_readcyclecounter();\n"
                    f"    if (lane == 0) {{ cyc[blockIdx.x * 8 + wave] = t1 - t0; cnt[blockIdx.x * 8 + wave] = n; }}\n  }}\n")
        else:
            src += "  } else {\n" + loop_a.replace(body_a, body_b) + "  }\n"
    src += POST + "}\n"
    return src


def variants():
    r"""[(name, label, nthreads, body_a, body_b, until_flag, mfma_per_iter_a, fillers_per_iter_b)]"""
    out = []
    n = [0]

    def add(label, nthreads, ba, bb=None, until=False, fb=0):
        out.append((f"k{n[0]}", label, nthreads, ba, bb, until, U, fb))
        n[0] += 1

    for order in ("rr", "chain4", "one"):
        add(f"A  1 wave/SIMD  MFMA only, order {order}", 256, body(order, None, 0))
    kinds = ["v_add", "v_fma", "v_pk_add", "v_pk_fma", "v_mov", "s_nop", "ds_read_b128", "ds_write_b64", "buf_load_x2"]
    for order in ("rr", "chain4"):
        for kind in kinds:
            for k in (1, 2, 4, 6, 8, 12):
                add(f"B  1 wave/SIMD  order {order:6s} + {k:2d} x {kind} per gap", 256, body(order, kind, k))
    for order in ("rr", "chain4"):
        add(f"D  2 waves/SIMD both MFMA only, order {order}", 512, body(order, None, 0), body(order, None, 0), False)
        for kind in ("v_add", "v_pk_add", "ds_read_b128", "buf_load_x2", "ds_write_b64"):
            for k in (1, 2, 4, 6):
                add(f"D  2 waves/SIMD both order {order:6s} + {k} x {kind} per gap", 512, body(order, kind, k), body(order, kind, k), False)
    # C: partner wave = fillers only, diluted with s_nop to several densities
    for kind in ("v_add", "v_pk_add", "buf_load_x2"):
        for fill, nops in ((1, 15), (4, 12), (16, 0)):
            lines = []
            for i in range(8):  # 8 groups of (fill fillers + nops s_nop)
                lines += [filler(kind, i * fill + j) for j in range(fill)] + ["s_nop 0"] * nops
            if kind == "ds_read_b128":
                lines.append("s_waitcnt lgkmcnt(8)")
            if kind == "buf_load_x2":
                lines.append("s_waitcnt vmcnt(8)")
            add(f"C  2 waves/SIMD wave w MFMA rr | partner {fill:2d} x {kind} + {nops:2d} s_nop", 512, body("rr", None, 0),
                "\\n\\t".join(lines), True, 8 * fill)
    # C2: the same pairing, but the MFMA wave has a real stall in it (one ds_read + s_waitcnt lgkmcnt(0) per 16 MFMAs)
    for kind in ("v_add", "v_pk_add"):
        for fill, nops in ((1, 15), (4, 12), (16, 0)):
            lines = []
            for i in range(8):
                lines += [filler(kind, i * fill + j) for j in range(fill)] + ["s_nop 0"] * nops
            add(f"C2 2 waves/SIMD wave w MFMA rr + 1 LDS round trip | partner {fill:2d} x {kind} + {nops:2d} s_nop", 512,
                body("rr", None, 0) + "\\n\\tds_read_b128 %20, %26\\n\\ts_waitcnt lgkmcnt(0)", "\\n\\t".join(lines), True, 8 * fill)
    return out


MAIN = r"""
int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 1500;
  float* d; unsigned long long* cyc; unsigned* cnt; float* g;
  (void)hipMalloc(&d, 1 << 16); (void)hipMalloc(&cyc, 256 * 8 * 8); (void)hipMalloc(&cnt, 256 * 8 * 4); (void)hipMalloc(&g, 1 << 20);
  (void)hipMemset(g, 0, 1 << 20);
  std::vector<unsigned long long> hc(256 * 8); std::vector<unsigned> hn(256 * 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("# v_mfma_f32_32x32x2_f32 beside other work on gfx950: 256 workgroups (one per CU), %d iterations x 16 MFMAs per wave\n", iters);
  printf("# cyc/MFMA = s_memtime cycles of the measured wave / its MFMA count (median over the chip; min in brackets); ideal = 64 per wave alone,\n");
  printf("# 128 per wave when two MFMA waves share a SIMD (= 64 per SIMD).  wall = hipEvent time of the launch.\n");
  for (const V& v : VARS) {
    (void)hipMemset(cyc, 0, 256 * 8 * 8); (void)hipMemset(cnt, 0, 256 * 8 * 4);
    v.fn<<<256, v.nthreads, 16384 * 4 + 64>>>(d, iters / 10 + 1, cyc, cnt, g);  // warm
    (void)hipEventRecord(e0);
    v.fn<<<256, v.nthreads, 16384 * 4 + 64>>>(d, iters, cyc, cnt, g);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(hc.data(), cyc, 256 * 8 * 8, hipMemcpyDeviceToHost); (void)hipMemcpy(hn.data(), cnt, 256 * 8 * 4, hipMemcpyDeviceToHost);
    std::vector<double> a, b;
    for (int blk = 0; blk < 256; ++blk) for (int w = 0; w < 4; ++w) a.push_back((double)hc[blk * 8 + w] / ((double)iters * v.mfma));
    std::sort(a.begin(), a.end());
    printf("%-78s cyc/MFMA %7.1f [%7.1f]  wall %8.1f us", v.label, a[a.size() / 2], a[0], ms * 1e3);
    if (v.nthreads == 512 && v.until) {
      for (int blk = 0; blk < 256; ++blk) for (int w = 4; w < 8; ++w)
        if (hc[blk * 8 + w]) b.push_back((double)hn[blk * 8 + w] * v.fb / (double)hc[blk * 8 + w] * 64.0);
      std::sort(b.begin(), b.end());
      if (!b.empty()) printf("  partner issued %6.2f fillers per 64 cycles", b[b.size() / 2]);
    } else if (v.nthreads == 512) {
      for (int blk = 0; blk < 256; ++blk) for (int w = 4; w < 8; ++w) b.push_back((double)hc[blk * 8 + w] / ((double)iters * v.mfma));
      std::sort(b.begin(), b.end());
      printf("  partner cyc/MFMA %7.1f", b[b.size() / 2]);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
"""


def generate() -> str:
    vs = variants()
    src = ("// generated by tools/mfma_overlap.py -- do not edit\n#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstdlib>\n"
           "#include <vector>\n#include <algorithm>\n"
           "typedef float f32x16 __attribute__((ext_vector_type(16)));\ntypedef float f32x2 __attribute__((ext_vector_type(2)));\n")
    for name, label, nt, ba, bb, until, mf, fb in vs:
        src += kernel(name, nt, ba, bb, until)
    src += ("struct V { const char* label; void (*fn)(float*, int, unsigned long long*, unsigned*, const float*); int nthreads; bool until; int mfma; int fb; };\n"
            "static const V VARS[] = {\n")
    for name, label, nt, ba, bb, until, mf, fb in vs:
        src += f'  {{"{label}", {name}, {nt}, {"true" if until else "false"}, {mf}, {fb}}},\n'
    src += "};\n" + MAIN
    return src


def build() -> str:
    os.makedirs(BUILD, exist_ok=True)
    path = os.path.join(BUILD, "mfma_overlap.hip")
    with open(path, "w") as f:
        f.write(generate())
    exe = os.path.join(BUILD, "mfma_overlap")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", exe, path])
    return exe


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "run"
    if cmd == "build":
        print(build())
    else:
        exe = os.path.join(BUILD, "mfma_overlap")
        if not os.path.exists(exe):
            build()
        sys.exit(subprocess.call([exe, *sys.argv[2:]]))
