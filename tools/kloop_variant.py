r"""A/B builds of the hand-scheduled K-loop streams:  python tools/kloop_variant.py NAME KEY=VALUE ... [-- -DMACRO ...]
generates variants of wino_kloop.inc / igemm_kloop.inc with the generators' overrides (KL_* / KG_*, honoured because this tool
sets AZ_KLOOP_AB=1) into azula_amd/csrc/_ab/inc_NAME/ -- the committed .inc files are never touched -- and builds
azula_amd/csrc/_ab/libazula_amd_NAME.so against them (tools/ab_build.py; select with AZULA_AMD_LIB=<path>)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
name = args[0]
defs = args[args.index("--") + 1:] if "--" in args else []
kv = [a for a in (args[1:args.index("--")] if "--" in args else args[1:])]
env = dict(os.environ, AZ_KLOOP_AB="1")
env.update(dict(a.split("=", 1) for a in kv))
csrc = os.path.join(ROOT, "azula_amd", "csrc")
inc_dir = os.path.join(csrc, "_ab", "inc_" + name)
os.makedirs(inc_dir, exist_ok=True)
for gen, inc, macro in (("gen_wino_kloop.py", "wino_kloop.inc", "AZ_WINO_KLOOP_INC"), ("gen_igemm_kloop.py", "igemm_kloop.inc", "AZ_IGEMM_KLOOP_INC")):
    out = os.path.join(inc_dir, inc)
    subprocess.run([sys.executable, os.path.join(csrc, gen), "--out", out], check=True, env=env, stdout=subprocess.DEVNULL)
    defs.append(f'-D{macro}="{out}"')
subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_build.py"), name, *defs], check=True)
