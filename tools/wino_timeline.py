r"""Per-workgroup timeline of the Winograd kernel (experiment build: `python tools/ab_build.py tl -DAZ_WINO_TL`).

    AZULA_AMD_LIB=azula_amd/csrc/_ab/libazula_amd_tl.so python tools/wino_timeline.py B H W Cin Cout

Stamps (s_memtime, waves 0 = gather role and 4 = filter role): 0 entry, 1 addresses done (before the K loop), 2 K loop done,
3 output transform + LDS exchange done, 4 epilogue rows read back, 5 stores issued, 6 stores acknowledged; word 7 = XCC id /
HW id.  Prints the median duration of each section and the gap between consecutive workgroups on one CU."""
import ctypes as C
import os
import statistics as st
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from azula_amd import _lib
from azula_amd.engine import Act, Builder

B, H, W, Cin, Cout = (int(v) for v in sys.argv[1:6])
dev = torch.device("cuda")
torch.manual_seed(0)
bld = Builder(dev)
x = Act(torch.randn(B * H * W * Cin, device=dev), B, H, W, Cin, Cin, True)
w = torch.randn(Cout, Cin, 3, 3, device=dev) / (Cin * 9) ** 0.5
y = bld.conv(x, bld.pack_conv(w, torch.randn(Cout, device=dev)), Cout, act=1, winograd=True)
bld.finish()
for _ in range(3):
    bld.tape.run()
torch.cuda.synchronize()
nwg = ((B * ((H + 1) // 2) * ((W + 1) // 2) + 63) // 64) * ((Cout + 63) // 64)
n = min(nwg, 8192)
buf = (C.c_ulonglong * (n * 16))()
fn = _lib.lib().az_debug_wino_timeline
fn.argtypes = [C.c_void_p, C.c_int]
assert fn(buf, n * 16) == 0
rows = [[buf[(i * 2 + r) * 8 + k] for k in range(8)] for i in range(n) for r in range(2)]
names = ["entry->addresses", "K loop", "out transform + exchange", "rows read back", "stores issued", "stores acked"]
for role in (0, 1):
    rr = rows[role::2]
    print("role", "gather (wave 0)" if role == 0 else "filter (wave 4)")
    for k, nm in enumerate(names):
        d = [r[k + 1] - r[k] for r in rr if r[k + 1] > r[k]]
        print(f"  {nm:28s} median {st.median(d):9.0f}  p10 {sorted(d)[len(d) // 10]:9.0f}  p90 {sorted(d)[9 * len(d) // 10]:9.0f} cycles")
    tot = [r[6] - r[0] for r in rr]
    print(f"  {'whole workgroup':28s} median {st.median(tot):9.0f}")
# consecutive workgroups of one CU: key = (xcc, hw id without the wave / simd bits)
by_cu = {}
for i in range(n):
    r = rows[2 * i]
    hw = r[7] & 0xffffffff
    key = (r[7] >> 32, (hw >> 8) & 0xf, (hw >> 12) & 0x1, (hw >> 13) & 0x7)  # cu_id, sh_id, se_id
    by_cu.setdefault(key, []).append((r[0], r[6]))
gaps = []
for key, v in by_cu.items():
    v.sort()
    gaps += [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
print(f"CUs seen: {len(by_cu)}; workgroups per CU: {n / max(1, len(by_cu)):.1f}")
if gaps:
    gaps.sort()
    print(f"gap end(stores acked) -> next workgroup's entry on the same CU: median {st.median(gaps):.0f}  p10 {gaps[len(gaps) // 10]:.0f}  p90 {gaps[9 * len(gaps) // 10]:.0f} cycles")
t0 = min(r[0] for r in rows)
t1 = max(r[6] for r in rows)
print(f"kernel span {t1 - t0} cycles (s_memtime)")
